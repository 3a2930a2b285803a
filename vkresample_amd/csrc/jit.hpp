// jit.hpp / jit.cpp -- run-time specialised plans (host side).
//
// The reference treats every 2*3*5*7-smooth size as first class because VkFFT GENERATES its shaders for the requested
// size at plan time and compiles them with glslang (vkFFT.h:4707-5189 scheduler, :6200-7700 generator, VkResample.cpp
// links glslang for it).  The MI355X-native counterpart: the register-resident kernels of kernels_pow2.hpp /
// kernels_mixed.hpp are C++ templates over the size, its radix factorisation and the upscale factor; for a plan with an
// integer or half-integer factor whose size has no ahead-of-time instantiation, fftup_plan_create
//   * picks factorizations (choose(): measured rules, a built-in table of tuner results for the MI355X, the user's
//     wisdom file),
//   * writes two ten-line translation units that name the instantiations (make_source(): row + column kernels, and
//     fused C2R+sharpen + stand-alone C2R, which depend on the output row length only and are shared between heights),
//   * compiles it for the plan's device with hipRTC (the ROCm run-time compiler, loaded with dlopen -- no link-time
//     dependency, and without it the plan silently stays on the size-generic kernels),
//   * loads the code object, relaxes the fused kernel's register bound if it spills (load()), and launches the kernels
//     from it (launch()).
// Code objects are cached in memory and on disk ($FFTUP_CACHE_DIR, else $XDG_CACHE_HOME/fftup, else ~/.cache/fftup),
// keyed by a hash of the translation unit, the instantiated names, the compiler options, the hipRTC version and the
// kernel headers' text.
// With FFTUP_FLAG_TUNE_PLAN the plan is also timed with the alternative factorizations of its dominant kernel
// (fused_candidates(), tune_fused() in fftup_plan.hip) and the decision kept in <cache dir>/wisdom.txt.
//
// The kernel headers' text is embedded in the library at build time (kernel_sources.inc, written by
// __graft_entry__.build() from the very files the ahead-of-time kernels are compiled from) and handed to hipRTC as
// in-memory headers; $FFTUP_KERNEL_DIR overrides it with a directory (development), and a library built without the
// generated file reads csrc/ next to libfftup.so.
//
// Environment: FFTUP_JIT=0 off; FFTUP_JIT_VERBOSE=1 says why a plan fell back and what the tuner measured;
// FFTUP_HIPRTC_LIB=<libhiprtc.so>.  Experiments and tests (FFTUP_EXPERIMENT="key=value;..", see experiment()): jit_row / jit_col /
// jit_coli / jit_fused ("r0,r1,.." resp. "threads:r0,r1,..") pin a factorization; jit_fused_opt="waves,ring" pins the fused
// kernel's register bound and ring-row placement; jit_no_builtin_wisdom; jit_dump=<file> writes the translation unit.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace fftup_jit {

struct Choice {
    int W = 0, H = 0, UW = 0;
    int U = 2;                         // integer upscale factor; 1 = half-integer factor D/2 (one spectrum buffer with all rows)
    int D = 4;                         // 2 DD x upscale factor: the spectrum rows hold kx = 0..UW DD / D
    int DD = 1;                        // 1: integer and half-integer factors (D = 2u); 2: quarter-integer ones (D = 4u, odd: -u 1.25 -> 5)
    int UH = 0;
    bool half = false;
    int row_kind = -1;                 // 0: k_row_r2c_t<W> (power of two, 8 points per thread); 1: k_row_r2c_m (three stages); 2: k_row_r2c (not specialised); 3: k_row_r2c_n
    int rr[3] = {0, 0, 0}, row_t = 0;
    std::vector<int> rn, cn;           // kind 3: k_row_r2c_n / k_col_n (N stages)
    std::vector<int> ci;               // col kind 5: the inverse transform of length UH
    int col_cols = 4;                  // col kinds 3-5: columns of a 4-wide spectrum tile per workgroup (4, or 2 for long columns)
    int col_kind = -1;                 // 0: k_col_t<H>; 1: k_col_m (three stages); 3: k_col_n (N stages); 4: k_col_u (N stages, U - 1 residues); 5: k_col_pad (half-integer factors)
    int cr[3] = {0, 0, 0}, col_tpc = 0;
    int fused_kind = -1;               // 0: FusedPlanPow2<UW>; 1: FusedPlanMr16<UW, UW/256>; 2: FusedPlanN<UW, T, 2, radices...>
    std::vector<int> fr;
    int fused_t = 0;
    int fused_wpe = 0;                 // kind 2: waves per SIMD the fused kernel's register allocation must allow
    bool fused_rr = true;              // kind 2: ring rows in registers
    std::vector<int> ct;               // stand-alone C2R (pre-sharpen tap): CtPlan<UW, ct_t, radices...>
    int ct_t = 0;
    // launch geometry derived from the above (what the device-side templates compute for themselves)
    int row_block = 0, col_block = 0;
    size_t col_lds = 0, fused_lds = 0, ct_lds = 0;
    bool u8out = false;                // fused kernel stores interleaved 8-bit RGB (FFTUP_FLAG_FUSE_U8_STORE)
};

// Test knobs: FFTUP_EXPERIMENT="key=value;key=value" pins factorizations and strip lengths for tests and tools (keys: aot,
// g_per_cu, pairs_per_strip, jit_tune, jit_row, jit_col, jit_coli, jit_fused, jit_fused_opt, jit_row_nstage, jit_col_nstage,
// jit_no_builtin_wisdom, jit_dump, jit_threads).  The parser exists only in a library built with -DFFTUP_TEST_KNOBS
// (libfftup_knobs.so, which the tests that pin something load); in the shipping library this returns nullptr, whatever the
// environment says.  Returns the value ("" for a bare key) or nullptr.
const char* experiment(const char* key);

struct FusedCand { int T; std::vector<int> r; };
enum { K_ROW_PLANAR = 0, K_ROW_U8, K_COL, K_FUSED, K_C2R_CT, K_COUNT };

// a compiled translation unit: code object + the lowered names of its kernels
struct Binary {
    std::string code;
    std::string lowered[K_COUNT];
};

// a code object loaded on one device
struct Module {
    hipModule_t mod[2] = {nullptr, nullptr};       // part 0: row + column kernels, part 1: fused C2R+sharpen + stand-alone C2R
    hipFunction_t fn[K_COUNT] = {};
    Choice choice;
    ~Module() { for (hipModule_t m : mod) if (m) (void)hipModuleUnload(m); }
};

// factorizations for a W x H plan with upscale factor D/2 (radices of the stand-alone C2R given); `arch`: the wisdom key's
// device part ("" = none); false: no specialised factorization exists
bool choose(int W, int H, int D, bool half, const std::vector<int>& ct_radices, Choice& c, const std::string& arch = "", bool use_wisdom = true, int DD = 1);
std::string describe(const Choice& c);
// alternatives for the fused kernel of a row length (the tuner's candidates), the chooser's pick first
std::vector<FusedCand> fused_candidates(int n, int D, size_t max, int DD = 1);
void set_fused_n(Choice& c, int T, const std::vector<int>& radices);
std::string fused_key(const Choice& c, const std::string& arch);
std::string fused_value(const Choice& c);
bool wisdom_lookup(const std::string& key, std::string& value);
void wisdom_store(const std::string& key, const std::string& value);
// both translation units of a plan compiled for `arch` (cached in memory and on disk)
bool compile_both(const Choice& c, const std::string& arch, Binary b[2], std::string& err);
// ... and loaded on the current device; relaxes the fused kernel's register bound if it spills.  nullptr: err says why
Module* load(Choice c, const std::string& arch, std::string& err);

template <class Params>
inline hipError_t launch(hipFunction_t f, dim3 grid, dim3 block, size_t lds, hipStream_t st, Params p)
{
    void* args[] = {&p};
    return hipModuleLaunchKernel(f, grid.x, grid.y, grid.z, block.x, block.y, block.z, (unsigned)lds, st, args, nullptr);
}

}  // namespace fftup_jit
