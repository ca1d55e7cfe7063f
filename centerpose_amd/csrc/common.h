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
