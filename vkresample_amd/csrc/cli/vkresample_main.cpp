// vkresample -- command-line front end of the MI355X FFT upscaler; drop-in for the VkResample binary.
//
// Mirrors the reference's process entry and per-thread pipeline (VkResample.cpp "VR"):
//   main()            VR:1795-1977   same flags, defaults, messages and exit codes
//   launchResample()  VR:1280-1780   plan once per thread, then per file: PNG decode -> upload ->
//                                    upscale x numIter -> download -> PNG encode
// All compute goes through the C ABI of include/fftup.h (HIP); PNG I/O is pngio (zlib).
// Extensions (do not change the reference behaviour when absent):
//   -alldevices   batched mode: thread t uses device (d + t) % device_count instead of all threads on -d
//   -fuseu8       the row kernel reads the uint8 image directly (README.md:31 roadmap item)
//   -fuseu8out    the last kernel stores the 8-bit image itself (FFTUP_FLAG_FUSE_U8_STORE): no float planes, no conversion launch
//   -wrapu8       u8 store wraps like the reference's C cast instead of saturating
//   -workqueue    batched mode: threads take the next unprocessed file from ONE shared counter (dynamic balancing over
//                 threads / GPUs of unequal speed) instead of the static stripe t+1, t+1+T, .. of VR:1622-1629
//   -tune         FFTUP_FLAG_TUNE_PLAN: time the alternatives for a size specialised at plan time, keep the fastest (wisdom file)
//   -overlap      FFTUP_FLAG_OVERLAP_ITERATIONS: the -n iterations alternate on the plan's streams ("Time:" = throughput; the
//                 default keeps them in order like the reference's barriers, VR:1217, vkFFT.h:7678)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "fftup.h"
#include "png_codec.hpp"

struct ResampleConfiguration {           // VkResampleConfiguration, VR:45-59
    const char* png_input_name = nullptr;
    const char* png_output_name = nullptr;
    float upscale = 1;
    uint32_t precision = 0;
    uint32_t numIter = 1;
    int device_id = 0;
    uint32_t fileUpload = 0;
    const char* ifolder_prefix = nullptr;
    const char* ofolder_prefix = nullptr;
    int numFiles = 0;
    int numThreads = 1;
    int threadId = 0;
    float sharpenConst = 0.2f;
    bool allDevices = false;
    uint32_t flags = 0;
    std::atomic<int>* workQueue = nullptr;   // -workqueue: next file number - 1, shared by all threads
    bool stageTimes = false;                 // -stagetimes: per-thread host time by stage (batched mode)
    bool gpuPng = false;                     // -gpupng: batched mode: the GPU delivers the finished PNG (fftup_submit_png)
};

static bool findFlag(char** start, char** end, const std::string& flag)      // VR:1782-1784: exact token match
{
    return std::find_if(start, end, [&](char* a) { return flag == a; }) != end;
}
static char* getFlagValue(char** start, char** end, const std::string& flag)  // VR:1785-1794
{
    char** value = std::find_if(start, end, [&](char* a) { return flag == a; });
    if (value == end) return nullptr;
    value++;
    return value != end ? *value : nullptr;
}

static int devices_list()                                                     // VR:239-268
{
    const int n = fftup_device_count();
    for (int i = 0; i < n; i++) {
        char name[256] = "";
        fftup_device_name(i, name, sizeof name);
        printf("Device id: %d name: %s API:HIP\n", i, name);
    }
    return n > 0 ? 0 : FFTUP_E_NO_DEVICE;
}

// Batched mode: the host threads of one GPU share ONE plan.  The reference gives every thread its own application (VR:1959-1969)
// because a Vulkan queue submission blocks its thread; here a thread's share of the GPU work is one asynchronous
// fftup_submit_rgb8 per file (0.5 ms of device time against ~250 ms of PNG decode + encode), and plan creation is serialised
// by the HIP runtime (~20 ms each, tools/plan_create_time.py): 64 plans for 64 threads cost 2.6 s of a 3.7 s job
// (profiles/r04_zi_cli_batch.txt).  Plans are keyed by (device, width, height) -- threads whose first file has another size
// get another plan, as in the reference -- and live until main() has joined the threads.
static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct SharedPlan { fftup_plan* plan = nullptr; fftup_info info{}; };
static std::mutex g_plans_mu;
static std::map<std::tuple<int, int, int>, SharedPlan> g_plans;
static int shared_plan(const fftup_config& cfg, SharedPlan* out)
{
    std::lock_guard<std::mutex> lock(g_plans_mu);
    SharedPlan& sp = g_plans[std::make_tuple((int)cfg.device, (int)cfg.width, (int)cfg.height)];
    if (!sp.plan) {
        const int res = fftup_plan_create(&sp.plan, &cfg);
        if (res != FFTUP_OK) { sp.plan = nullptr; return res; }
        fftup_plan_info(sp.plan, &sp.info);
    }
    *out = sp;
    return FFTUP_OK;
}
static void shared_plans_destroy()
{
    std::lock_guard<std::mutex> lock(g_plans_mu);
    for (auto& kv : g_plans)
        if (kv.second.plan) { fftup_drain(kv.second.plan); fftup_plan_destroy(kv.second.plan); }
    g_plans.clear();
}

static int launchResample(ResampleConfiguration config)                      // VR:1280-1780
{
    if (config.threadId == 0) printf("VkResample - FFT based upscaling\n");
    const double tStart = now_ms();
    double tDecode = 0, tSubmit = 0, tWait = 0, tEncode = 0;   // -stagetimes
    // which file (1-based number) is this thread's f-th one, 0 = none left.  Static stripe of the reference: f*T + t + 1 for
    // f < numLocalFiles (VR:1622-1629); -workqueue: whatever the shared counter hands out next.
    int numLocalFiles = 1;
    if (config.fileUpload) {                                                   // VR:1622-1625
        numLocalFiles = (int)std::ceil(config.numFiles / (float)config.numThreads);
        if ((numLocalFiles - 1) * config.numThreads + config.threadId > config.numFiles - 1) numLocalFiles--;
    }
    std::vector<int> claimed;
    auto fileAt = [&](int f) -> int {
        while ((int)claimed.size() <= f) {
            int n;
            if (config.workQueue) { n = config.workQueue->fetch_add(1) + 1; if (n > config.numFiles) n = 0; }
            else n = ((int)claimed.size() < numLocalFiles || claimed.empty()) ? (int)claimed.size() * config.numThreads + config.threadId + 1 : 0;
            claimed.push_back(n);
        }
        // static stripe with more threads than files: the reference still loads file threadId + 1 for the plan's size but
        // processes nothing (VR:1622-1629) -- the first claim names that file, the loops see none
        if (!config.workQueue && config.fileUpload && f >= numLocalFiles) return 0;
        return claimed[(size_t)f];
    };
    char fileName[1024];
    if (config.fileUpload) {
        if (config.workQueue && fileAt(0) == 0) {                              // -workqueue with more threads than files
            printf("Thread %d finished. No files left in the queue\n", config.threadId);
            return FFTUP_OK;
        }
        // (static stripe: the size comes from file threadId + 1 even when the thread has no file to process, VR:1357, 1622-1629)
        snprintf(fileName, sizeof fileName, "%s/%06d.png", config.ifolder_prefix, config.workQueue ? fileAt(0) : config.threadId + 1);   // VR:1357
    }
    else snprintf(fileName, sizeof fileName, "%s", config.png_input_name);

    std::vector<uint8_t> png_input;
    int width = 0, height = 0, channels = 0;
    std::string err;
    if (!pngio::load_rgb8(fileName, png_input, width, height, channels, err)) {
        printf("Image not found\n");                                           // VR:1364-1367
        return FFTUP_E_INCOMPLETE;
    }
    tDecode += now_ms() - tStart;
    int device = config.device_id;
    if (config.allDevices) {
        const int nd = fftup_device_count();
        if (nd > 0) device = (config.device_id + config.threadId) % nd;
    }
    fftup_config cfg{};
    cfg.width = (uint32_t)width; cfg.height = (uint32_t)height; cfg.channels = 3;
    cfg.upscale = config.upscale; cfg.precision = config.precision; cfg.sharpen = config.sharpenConst;
    cfg.device = device; cfg.flags = config.flags; cfg.ring = config.fileUpload ? 2 : 1;
    // -n N: the N iterations run in order on one stream, as the reference's one command buffer with its barriers does
    // (VR:1260-1265): "Time:" is the figure comparable with the reference's, and the plan (strip cuts included) is the same for
    // every N.  -overlap (extension, config.flags): the iterations alternate on the plan's streams -- a throughput figure.
    // the batched path below runs on the GPU's shared plan: one frame in flight per thread, sixteen slots at most
    // (a slot is busy for ~0.5 ms per frame; a thread comes back after tens of ms of codec work)
    const bool streamed = config.fileUpload && config.numIter == 1 && config.numFiles > 1;
    int sharers = config.numThreads;
    if (config.allDevices && fftup_device_count() > 0) sharers = (config.numThreads + fftup_device_count() - 1) / fftup_device_count();
    if (streamed) cfg.ring = (uint32_t)std::min(16, std::max(2, sharers));
    fftup_plan* plan = nullptr;
    fftup_info info{};
    int res;
    if (streamed) {
        SharedPlan sp;
        res = shared_plan(cfg, &sp);
        plan = sp.plan; info = sp.info;
    } else {
        res = fftup_plan_create(&plan, &cfg);
        if (res == FFTUP_OK) fftup_plan_info(plan, &info);
    }
    if (res != FFTUP_OK) {
        printf("Plan creation failed: %s (%s)\n", fftup_strerror(res), fftup_last_error());
        return res;
    }
    if (config.threadId == 0) {
        const int mb = (int)(info.device_bytes / 1024 / 1024);                 // VR:1450 (a shared plan is counted once per GPU)
        if (streamed) printf("VRAM per thread: %d MB Total: %d MB\n", mb / sharers, mb * ((config.numThreads + sharers - 1) / sharers));
        else printf("VRAM per thread: %d MB Total: %d MB\n", mb, config.numThreads * mb);
    }
    const uint32_t uW = info.out_width, uH = info.out_height;
    std::vector<uint8_t> png_output(streamed ? 0 : (size_t)uW * uH * 3);

    if (streamed) {
        // batched mode: the frame travels through the GPU's shared plan as ONE asynchronous submission (H2D,
        // kernels, D2H from/to this thread's page-locked buffers), ~1 ms between submit and wait; the thread's time is the
        // PNG codec (tens of ms per file), so the overlap that matters is between the THREADS -- one frame and one pair of
        // buffers per thread (a second pair would hide 1 ms per file and double the page-locking, which the driver serialises).
        const size_t inBytes = (size_t)width * height * 3, outBytes = (size_t)uW * uH * 3;
        const bool gpuPng = config.gpuPng && config.precision != 1;    // (-p 1: the host encodes)
        const size_t pngCap = gpuPng ? fftup_png_bound(plan) : 0;
        uint8_t* pin = (uint8_t*)fftup_host_alloc(inBytes);
        uint8_t* pout = (uint8_t*)fftup_host_alloc(gpuPng ? pngCap : outBytes);
        auto release = [&]() { fftup_host_free(pin); fftup_host_free(pout); };     // (no frame of this thread is in flight here)
        if (!pin || !pout) {
            printf("Upscale failed: %s (%s)\n", fftup_strerror(FFTUP_E_OUT_OF_MEMORY), fftup_last_error());
            release();
            return FFTUP_E_OUT_OF_MEMORY;
        }
        const double tSetup = now_ms() - tStart - tDecode;             // plan (shared: the first thread creates it) + page-locked buffers
        int f = 0;
        for (; fileAt(f) > 0; f++) {
            double t0 = now_ms();
            if (f > 0) {
                snprintf(fileName, sizeof fileName, "%s/%06d.png", config.ifolder_prefix, fileAt(f));
                int w2 = 0, h2 = 0;
                if (!pngio::load_rgb8(fileName, png_input, w2, h2, channels, err) || w2 != width || h2 != height) {
                    printf("Image not found\n");                               // VR:1631-1634
                    release();
                    return FFTUP_E_INCOMPLETE;
                }
                tDecode += now_ms() - t0;
                t0 = now_ms();
            }
            memcpy(pin, png_input.data(), inBytes);
            uint64_t ticket = 0;
            res = gpuPng ? fftup_submit_png(plan, pin, (size_t)width * 3, pout, pngCap, &ticket)
                         : fftup_submit_rgb8(plan, pin, (size_t)width * 3, pout, (size_t)uW * 3, &ticket);
            if (res != FFTUP_OK) {
                printf("Upscale failed: %s (%s)\n", fftup_strerror(res), fftup_last_error());
                release();
                return res;
            }
            double t1 = now_ms();
            tSubmit += t1 - t0;
            size_t pngBytes = 0;
            res = gpuPng ? fftup_wait_png(plan, ticket, pout, pngCap, &pngBytes) : fftup_wait(plan, ticket);
            t0 = now_ms();
            tWait += t0 - t1;
            if (res != FFTUP_OK) {
                printf("Download failed: %s (%s)\n", fftup_strerror(res), fftup_last_error());
                release();
                return res;
            }
            char outName[1024];
            snprintf(outName, sizeof outName, "%s/%06d.png", config.ofolder_prefix, fileAt(f));
            if (gpuPng) {                                           // the file arrived finished: filters, deflate, checksums
                FILE* fo = fopen(outName, "wb");
                const bool ok = fo && fwrite(pout, 1, pngBytes, fo) == pngBytes;
                if ((fo && fclose(fo) != 0) || !ok) printf("Could not write %s: write error\n", outName);
            } else if (!pngio::write_rgb8(outName, pout, (int)uW, (int)uH, (size_t)uW * 3, err))
                printf("Could not write %s: %s\n", outName, err.c_str());
            tEncode += now_ms() - t0;
        }
        const double tRel = now_ms();
        release();
        if (config.stageTimes)
            printf("Thread %d: %d files in %.0f ms: setup %.0f, decode %.0f, copy+submit %.0f, wait %.0f, encode+write %.0f, release %.0f ms\n",
                   config.threadId, f, now_ms() - tStart, tSetup, tDecode, tSubmit, tWait, tEncode, now_ms() - tRel);
        printf("Thread %d finished. Device name: %s API:HIP\n", config.threadId, info.device_name);   // VR:1773
        return FFTUP_OK;
    }
    for (int f = 0; config.fileUpload ? fileAt(f) > 0 : f < 1; f++) {
        if (f > 0) {
            snprintf(fileName, sizeof fileName, "%s/%06d.png", config.ifolder_prefix, fileAt(f));
            int w2 = 0, h2 = 0;
            if (!pngio::load_rgb8(fileName, png_input, w2, h2, channels, err) || w2 != width || h2 != height) {
                printf("Image not found\n");                                   // VR:1631-1634 (all files share one size)
                fftup_plan_destroy(plan);
                return FFTUP_E_INCOMPLETE;
            }
        }
        res = fftup_upload_rgb8(plan, png_input.data(), (size_t)width * 3);   // pack loop + transferDataFromCPU
        double totTime = 0;
        if (res == FFTUP_OK) res = fftup_execute(plan, config.numIter, &totTime);   // performVulkanUpscale, VR:1692
        if (res != FFTUP_OK) {
            printf("Upscale failed: %s (%s)\n", fftup_strerror(res), fftup_last_error());
            fftup_plan_destroy(plan);
            return res;
        }
        if (!config.fileUpload)
            printf("VkResample %0.1fx upscale: %dx%d to %dx%d Time: %0.3f ms\n", config.upscale, width, height, uW, uH, totTime);   // VR:1694
        res = fftup_download_rgb8(plan, 0, png_output.data(), (size_t)uW * 3);  // transferDataToCPU + unpack loop
        if (res != FFTUP_OK) {
            printf("Download failed: %s (%s)\n", fftup_strerror(res), fftup_last_error());
            fftup_plan_destroy(plan);
            return res;
        }
        char outName[1024];
        if (config.fileUpload) snprintf(outName, sizeof outName, "%s/%06d.png", config.ofolder_prefix, fileAt(f));
        else if (config.png_output_name) snprintf(outName, sizeof outName, "%s", config.png_output_name);
        else snprintf(outName, sizeof outName, "%d_%d_upscaled.png", width, (int)uW);   // VR:1706
        if (!pngio::write_rgb8(outName, png_output.data(), (int)uW, (int)uH, (size_t)uW * 3, err))
            printf("Could not write %s: %s\n", outName, err.c_str());
    }
    fftup_plan_destroy(plan);
    printf("Thread %d finished. Device name: %s API:HIP\n", config.threadId, info.device_name);   // VR:1773
    return FFTUP_OK;
}

int main(int argc, char* argv[])
{
    ResampleConfiguration config;
    char** B = argv;
    char** E = argv + argc;
    if (findFlag(B, E, "-h")) {
        // (first line: the reference's own banner, VR:1808, for scripts that look for it; then whose build this is)
        printf("VkResample v1.0.2 (16-01-2021). Author: Tolmachev Dmitrii\n");
        printf("(command line reproduced by the MI355X/HIP build, %s)\n", fftup_version());
        printf("PNG images only.\n");
        printf("	-h: this help\n");
        printf("	-devices: list the available GPUs\n");
        printf("	-d X: GPU to use (default 0)\n");
        printf("	-u X: upscale factor (float, or a ratio like 4/3; the upscaled sizes must factor into 2s, 3s, 5s and 7s)\n");
        printf("	-p X: specify precision (0 - single, 1 - double, 2 - half, default - single)\n");
        printf("	-s X: sharpening factor, 0.0-0.2 (default 0.2)\n");
        printf("	-n X: how many times to run the upscale; removes launch overhead from the reported time (default 1)\n");
        printf("Single image mode:\n");
        printf("	-i NAME: input png\n");
        printf("	-o NAME: output png (default <width>_<upscaled width>_upscaled.png)\n");
        printf("Batched mode:\n");
        printf("	-ifolder X: input folder; files are X/000001.png, X/000002.png, ...\n");
        printf("	-ofolder X: output folder, same naming\n");
        printf("	-numfiles X: number of images\n");
        printf("	-numthreads X: host threads, each with its own plan; thread t takes files t+1, t+1+X, ...\n");
        printf("Extensions:\n");
        printf("	-alldevices: thread t runs on GPU (d + t) %% count\n");
        printf("	-fuseu8: FFT kernel reads the 8-bit image directly\n");
        printf("	-fuseu8out: the last kernel writes the 8-bit image directly (no float planes, no conversion pass)\n");
        printf("	-wrapu8: 8-bit store wraps like the original's C cast instead of saturating\n");
        printf("	-workqueue: batched mode: threads take the next unprocessed file from one shared counter instead of the fixed stripe\n");
        printf("	-gpupng: batched mode: the GPU also encodes the PNG (row filters, Huffman-only deflate, Adler-32); the host writes the file\n");
        printf("	-stagetimes: batched mode: every thread reports its host time by stage (decode, submit, wait, encode)\n");
        printf("	-tune: sizes whose kernels are specialised at plan time: measure the alternatives once, remember the fastest\n");
        printf("	-overlap: the -n iterations overlap on several streams: 'Time:' becomes a throughput figure, not the original's serial one\n");
        return 0;
    }
    if (findFlag(B, E, "-devices")) return devices_list();
    if (findFlag(B, E, "-d")) {
        char* v = getFlagValue(B, E, "-d");
        if (v) sscanf(v, "%d", &config.device_id);
        else { printf("No device is selected with -d flag\n"); return 1; }
    }
    if (findFlag(B, E, "-n")) {
        char* v = getFlagValue(B, E, "-n");
        if (v) sscanf(v, "%u", &config.numIter);
        else { printf("No number is selected with -n flag\n"); return 1; }
        if (config.numIter == 0) config.numIter = 1;
    }
    if (findFlag(B, E, "-p")) {
        char* v = getFlagValue(B, E, "-p");
        if (v) sscanf(v, "%u", &config.precision);
        else { printf("No precision is selected with -p flag\n"); return 1; }
    }
    if (findFlag(B, E, "-s")) {
        char* v = getFlagValue(B, E, "-s");
        if (v) sscanf(v, "%f", &config.sharpenConst);
        else { printf("No sharpening parameter is selected with -s flag\n"); return 1; }
    }
    if (findFlag(B, E, "-u")) {
        char* v = getFlagValue(B, E, "-u");
        // (VR:1883: "%f".  Extension: "-u 4/3" -- a ratio, divided in float: the nearest float of 4/3 is what makes 1920 x 1080 come out
        // as exactly 2560 x 1440 in the reference's float arithmetic; typed as a decimal it takes eight digits, 1.3333334)
        float num = 0.f, den = 0.f;
        if (v && sscanf(v, "%f/%f", &num, &den) == 2 && den != 0.f) config.upscale = num / den;
        else if (v) sscanf(v, "%f", &config.upscale);
        else printf("No proper upscale factor is selected with -u flag, default 1\n");
    } else {
        printf("No upscale factor is selected with -u flag, default 1\n");
    }
    config.allDevices = findFlag(B, E, "-alldevices");
    if (findFlag(B, E, "-fuseu8")) config.flags |= FFTUP_FLAG_FUSE_U8_LOAD;
    if (findFlag(B, E, "-fuseu8out")) config.flags |= FFTUP_FLAG_FUSE_U8_STORE;
    if (findFlag(B, E, "-wrapu8")) config.flags |= FFTUP_FLAG_U8_WRAP;
    if (findFlag(B, E, "-tune")) config.flags |= FFTUP_FLAG_TUNE_PLAN;
    if (findFlag(B, E, "-overlap")) config.flags |= FFTUP_FLAG_OVERLAP_ITERATIONS;
    config.stageTimes = findFlag(B, E, "-stagetimes");
    config.gpuPng = findFlag(B, E, "-gpupng");

    if (!findFlag(B, E, "-ifolder")) {
        config.fileUpload = 0;
        config.png_input_name = getFlagValue(B, E, "-i");
        if (!config.png_input_name) { printf("No input file is selected with -i flag\n"); return 1; }
        if (findFlag(B, E, "-o")) {
            config.png_output_name = getFlagValue(B, E, "-o");
            if (!config.png_output_name) { printf("No output file is selected with -o flag\n"); return 1; }
        }
    } else {
        config.fileUpload = 1;
        config.ifolder_prefix = getFlagValue(B, E, "-ifolder");
        if (!config.ifolder_prefix) { printf("No input folder+prefix is selected with -ifolder flag\n"); return 1; }
        config.ofolder_prefix = getFlagValue(B, E, "-ofolder");
        if (!config.ofolder_prefix) { printf("No output folder+prefix is selected with -ofolder flag\n"); return 1; }
        if (findFlag(B, E, "-numthreads")) {
            char* v = getFlagValue(B, E, "-numthreads");
            if (v) sscanf(v, "%d", &config.numThreads);
            else { printf("No numThreads is selected with -numthreads flag\n"); return 1; }
        }
        char* v = getFlagValue(B, E, "-numfiles");                // the reference leaves numFiles uninitialised when
        if (v) sscanf(v, "%d", &config.numFiles);                  // the flag is missing (quirk B10): required here
        else { printf("No numFiles is selected with -numfiles flag\n"); return 1; }
        if (config.numThreads < 1) config.numThreads = 1;
        if (config.numFiles < 1) { printf("No numFiles is selected with -numfiles flag\n"); return 1; }
    }
    std::atomic<int> queue{0};
    if (config.fileUpload && findFlag(B, E, "-workqueue")) config.workQueue = &queue;
    auto timeSubmit = std::chrono::system_clock::now();
    std::vector<std::thread> threads;
    std::vector<int> results((size_t)config.numThreads, 0);
    for (int i = 0; i < config.numThreads; i++) {
        ResampleConfiguration loc = config;
        loc.threadId = i;
        threads.emplace_back([loc, i, &results]() { results[(size_t)i] = launchResample(loc); });
    }
    for (auto& t : threads) t.join();
    shared_plans_destroy();
    auto timeEnd = std::chrono::system_clock::now();
    double totTime = std::chrono::duration_cast<std::chrono::microseconds>(timeEnd - timeSubmit).count() * 0.001;
    printf("Total time: %0.3f s\n", totTime / 1000);
    // the reference returns VK_SUCCESS whatever its threads did (VR:1975); a failing thread is reported here
    for (int r : results) if (r != FFTUP_OK) return r;
    return 0;
}
