// Shared helpers for the centerpose MI355X (gfx950) kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CP_WAVE 64

// error plumbing for the C ABI (thread-local last error string)
extern "C" const char* cp_last_error(void);
void cp_set_error(const char* fmt, ...);
void cp_note_kernel(const char* fmt, ...);      // cp_last_kernel(): the kernel instantiation a launcher dispatched to

#define CP_CHECK_ARG(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            cp_set_error(__VA_ARGS__);       \
            return 1;                        \
        }                                    \
    } while (0)

#define CP_CHECK_LAUNCH(name)                                                   \
    do {                                                                        \
        hipError_t e_ = hipGetLastError();                                      \
        if (e_ != hipSuccess) {                                                 \
            cp_set_error("%s: kernel launch failed: %s", name, hipGetErrorString(e_)); \
            return 2;                                                           \
        }                                                                       \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline int cp_cdiv(int a, int b) { return (a + b - 1) / b; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE property of a kernel: one guard per (kernel instantiation,
// device).  `ensure(kernel, bytes)` raises the kernel's limit to `bytes` unless this device already has at least that much; the
// check and the attribute call sit under one mutex (cold path: taken once per instantiation and device in practice), so a second
// thread can never see the new size before the attribute is in place, and a failed call leaves the guard where it was -- the
// next launch tries again and reports the same error instead of launching with too much dynamic LDS.
#include <mutex>
struct CpLdsGuard {
    std::mutex mu;
    int have[16] = {};
    hipError_t ensure(const void* kernel, int bytes) {
        int d = 0;
        (void)hipGetDevice(&d);
        std::lock_guard<std::mutex> lock(mu);
        if (have[d & 15] >= bytes) return hipSuccess;
        const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) have[d & 15] = bytes;
        return e;
    }
};

// Compute units of the current device (persistent kernels size their grid as blocks-per-CU x CUs).  Queried once per device,
// thread-safe; 256 (MI355X) if the query fails.
static inline int cp_num_cus()
{
    static std::mutex mu;
    static int cus[16] = {};
    int d = 0;
    (void)hipGetDevice(&d);
    std::lock_guard<std::mutex> lock(mu);
    int& n = cus[d & 15];
    if (n <= 0 && (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0)) n = 256;
    return n;
}

// activation codes of the C ABI (include/centerpose_hip.h).  h-swish / h-sigmoid as the reference writes them
// (lib/models/backbones/mobilenet/mobilenetv3.py:87-96): x * relu6(x + 3) / 6 and relu6(x + 3) / 6, true division.
#define CP_ACT_NONE_ 0
#define CP_ACT_RELU_ 1
#define CP_ACT_SIGMOID_ 2
#define CP_ACT_HSWISH_ 3
#define CP_ACT_HSIGMOID_ 4
// ReLU as ONE v_max_f32.  fmaxf(v, 0.f) compiles to two (the compiler first quiets a possible signalling NaN with
// v_max_f32 v, v, v); PMC ranks the conv kernels by "other VALU per MFMA" (profiles/r3_mfma_util.json) and every VALU instruction
// takes issue cycles the matrix pipe then cannot use.  Same result for every input (max(0, NaN) = 0 either way).
__device__ __forceinline__ float cp_relu(float v)
{
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(v));
    return r;
}
typedef float cp_v2f __attribute__((ext_vector_type(2)));
// v * sc + sh on a float4 as two v_pk_fma_f32 (same bits as four scalar fmas)
__device__ __forceinline__ float4 cp_scale_shift4(float4 v, float4 sc, float4 sh)
{
    const cp_v2f lo = __builtin_elementwise_fma((cp_v2f){v.x, v.y}, (cp_v2f){sc.x, sc.y}, (cp_v2f){sh.x, sh.y});
    const cp_v2f hi = __builtin_elementwise_fma((cp_v2f){v.z, v.w}, (cp_v2f){sc.z, sc.w}, (cp_v2f){sh.z, sh.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 cp_relu4(float4 v) { return make_float4(cp_relu(v.x), cp_relu(v.y), cp_relu(v.z), cp_relu(v.w)); }

__device__ __forceinline__ float cp_act(float v, int act)
{
    if (act == CP_ACT_RELU_) return cp_relu(v);
    if (act == CP_ACT_SIGMOID_) return 1.0f / (1.0f + __expf(-v));
    if (act == CP_ACT_HSWISH_) return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f;
    if (act == CP_ACT_HSIGMOID_) return fminf(fmaxf(v + 3.f, 0.f), 6.f) / 6.f;
    return v;
}
