// The HIP-free part of common.h: what the pure host sources (gzhex.cpp, ptfile.cpp, config.cpp, host_threads.h) need — so that they build
// without a HIP toolchain in the sanitizer harness (tests/csrc/Makefile).
#pragma once
#include "../../include/mzgpu.h"
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

namespace mz {

void setError(const char* fmt, ...);
bool compressToHex(const uint8_t* data, size_t n, std::string* hex); // gzhex.cpp: utils::compressString (ref utils/utils.h:35-91)
const char* lastError();

} // namespace mz
