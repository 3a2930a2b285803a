// kernels_pow2.hpp -- size-specialised (compile-time plan) kernels for the headline sizes.
#pragma once
#include "fft_engine.hpp"
namespace fftup {
}  // namespace fftup
