// fbsmi_host.h -- host-side helpers shared by the translation units of libfbsmi.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace fbsmi {

// caller-provided scratch, carved by carve(): see fbsmi_workspace_bytes()
struct Workspace {
    float* part0;  // per-workgroup partial sums
    float* part1;  // per-workgroup partial maxima
    float* pref;   // top-level (P, E) per partial (only when the partials exceed one in-kernel tree)
    float* pend;
    float* scal;   // 64 scalars
    float* tmp0;   // n+1 floats
    float* tmp1;
    float* tmp2;
};

int fail(int code, const std::string& msg);
Workspace carve(void* ws, int64_t n);
int items_for(int64_t n);

#define FBSMI_HIP_TRY(call)                                                                         \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) return ::fbsmi::fail(-2, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)

}  // namespace fbsmi
