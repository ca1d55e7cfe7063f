// Micro-benchmark (stage A): ceiling of an fp32-EQUIVALENT GEMM inner loop on the bf16 matrix pipe of gfx950.
// Every fp32 operand is three bf16 terms (x = x1 + x2 + x3 exactly, 8+8+8 significand bits, round-to-nearest at each level); the
// six significant cross products x1y1, x1y2, x2y1, x1y3, x2y2, x3y1 go to v_mfma_f32_32x32x16_bf16 with fp32 accumulate
// (dropped terms <= 2^-24 relative).  Ceiling 2516 / 6 = 419 TF fp32-equivalent against 157.3 TF on the f32 MFMA.
// One iteration = one k-step (16) of a wave's 64 x 64 output tile: 4 tiles x 6 MFMAs.
//   MODE 0  register-fed (pipe ceiling)            MODE 1  + 12 ds_read_b128 (3 terms x (2 A + 2 B) fragments)
//   MODE 2  + split of 8 fresh fp32 activations per thread (v_cvt_pk_bf16_f32 / v_pk_add_f32) + 6 ds_write_b128 + barrier
//   MODE 3  + the k-step's global loads (2 float4 A + 3 uint4 pre-split B per thread, L2-resident)
//   MODE 4  as 3, software-pipelined: the split of the NEXT k-step's activations (loaded one iteration earlier) is interleaved with
//           the MFMAs (one split2 pair per output tile), the loads for the k-step after that are issued behind the LDS writes
//   MODE 6  (round 6, VERDICT r5 #3 "stage C" go / no-go) BOTH operands PRE-SPLIT in global memory ([row][k-step][3 terms][16 bf16]: the
//           producer's epilogue would write the activation that way, 6 B per element) and staged with global_load_lds_dwordx4
//           (LDS-DMA): no v_cvt_pk, no ds_write.  LDS image [term][row][32 B], the two 16-byte halves of a row XOR-swizzled on the
//           SOURCE address (the DMA destination is lane-linear) so that the ds_read_b128 fragment reads stay conflict-free; two LDS
//           buffers, the next k-step's 6 DMAs per thread issued in front of the MFMA block, one __syncthreads() (vmcnt(0)) per k-step.
//           kglds<.., 2>: two k-steps (K = 32) per barrier.  go: >= 330 TF fp32-equivalent.
// The last line is the f32 MFMA loop of mfma_peak.hip (LDS-fed, same process) = "the fp32 loop" of the kill criterion.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt2(float a, float b)
{
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split2(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3)
{
    p1 = cvt2(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, p1 << 16), r1 = x1 - __builtin_bit_cast(float, p1 & 0xffff0000u);
    p2 = cvt2(r0, r1);
    p3 = cvt2(r0 - __builtin_bit_cast(float, p2 << 16), r1 - __builtin_bit_cast(float, p2 & 0xffff0000u));
}
__device__ __forceinline__ f32x16 mm(v4u a, v4u b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#define LDR 28      // LDS row stride in dwords: 3 terms x 16 bf16 + 16 B pad (conflict-free ds_read_b128)

template <int MODE, int BPC>
__global__ __launch_bounds__(256, BPC) void k(float* out, const float* __restrict__ ga, const unsigned* __restrict__ gb, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * 256 * LDR];      // [buf][A 128 rows | B 128 rows][LDR]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 2 * 256 * LDR; i += 256) lds[i] = 0x3f803f80u + (i & 7);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int g = lane >> 5, il = lane & 31, wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
    v4u af[2][3], bf[2][3];
    for (int i = 0; i < 2; ++i) for (int t = 0; t < 3; ++t) { af[i][t] = (v4u){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; bf[i][t] = af[i][t]; }
    const float* pa = ga + (size_t)(blockIdx.x & 63) * 128 * 1024 + (tid >> 1) * 1024 + (tid & 1) * 8;
    const unsigned* pb = gb + tid * 4;
    v4f a0 = {1.f, 1.5f, 0.3f, 0.7f}, a1 = {0.1f, 0.9f, 2.3f, 0.01f};
    v4u b0 = {}, b1 = {}, b2 = {};
    int cur = 0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const unsigned* As = lds + cur * 256 * LDR;
        const unsigned* Bs = As + 128 * LDR;
        if (MODE >= 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    af[i][t] = *reinterpret_cast<const v4u*>(As + (wm0 + i * 32 + il) * LDR + t * 8 + g * 4);
                    bf[i][t] = *reinterpret_cast<const v4u*>(Bs + (wn0 + i * 32 + il) * LDR + t * 8 + g * 4);
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        unsigned q1[4], q2[4], q3[4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (MODE >= 4) {                               // a quarter of the next k-step's split beside this tile's six MFMAs
                    const int e = i * 2 + j;
                    const float x0 = e == 0 ? a0.x : e == 1 ? a0.z : e == 2 ? a1.x : a1.z, x1 = e == 0 ? a0.y : e == 1 ? a0.w : e == 2 ? a1.y : a1.w;
                    split2(x0, x1, q1[e], q2[e], q3[e]);
                }
                if (MODE == 3 && i == 1 && j == 0) {          // next k-step's loads from inside the MFMA block
                    const int ko = (it & 63) * 16;
                    a0 = *(const __attribute__((address_space(1))) v4f*)(pa + ko);
                    a1 = *(const __attribute__((address_space(1))) v4f*)(pa + ko + 4);
                    b0 = *(const __attribute__((address_space(1))) v4u*)(pb + (it & 63) * 3072);
                    b1 = *(const __attribute__((address_space(1))) v4u*)(pb + (it & 63) * 3072 + 1024);
                    b2 = *(const __attribute__((address_space(1))) v4u*)(pb + (it & 63) * 3072 + 2048);
                }
                acc[i][j] = mm(af[i][2], bf[j][0], acc[i][j]);
                acc[i][j] = mm(af[i][1], bf[j][1], acc[i][j]);
                acc[i][j] = mm(af[i][0], bf[j][2], acc[i][j]);
                acc[i][j] = mm(af[i][1], bf[j][0], acc[i][j]);
                acc[i][j] = mm(af[i][0], bf[j][1], acc[i][j]);
                acc[i][j] = mm(af[i][0], bf[j][0], acc[i][j]);
            }
        if (MODE < 4) __builtin_amdgcn_sched_barrier(0);
        if (MODE >= 2) {
            unsigned* An = lds + (cur ^ 1) * 256 * LDR;
            unsigned* Bn = An + 128 * LDR;
            if (MODE == 2) { a0 += 0.25f; a1 *= 1.0001f; b0 += 1u; }
            v4u t1, t2, t3, u1, u2, u3;
            if (MODE < 4) {
                split2(a0.x, a0.y, q1[0], q2[0], q3[0]); split2(a0.z, a0.w, q1[1], q2[1], q3[1]);
                split2(a1.x, a1.y, q1[2], q2[2], q3[2]); split2(a1.z, a1.w, q1[3], q2[3], q3[3]);
            }
            t1 = (v4u){q1[0], q1[1], q1[2], q1[3]}; t2 = (v4u){q2[0], q2[1], q2[2], q2[3]}; t3 = (v4u){q3[0], q3[1], q3[2], q3[3]};
            unsigned* ar = An + (tid >> 1) * LDR + (tid & 1) * 4;
            *reinterpret_cast<v4u*>(ar) = t1; *reinterpret_cast<v4u*>(ar + 8) = t2; *reinterpret_cast<v4u*>(ar + 16) = t3;
            u1 = b0; u2 = b1; u3 = b2;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int idx = tid + s * 256;
                *reinterpret_cast<v4u*>(Bn + (idx / 6) * LDR + (idx % 6) * 4) = s == 0 ? u1 : s == 1 ? u2 : u3;
            }
            if (MODE >= 4) {                                   // loads for the k-step after next: a whole iteration to arrive
                const int ko = (it & 63) * 16;
                a0 = *(const __attribute__((address_space(1))) v4f*)(pa + ko);
                a1 = *(const __attribute__((address_space(1))) v4f*)(pa + ko + 4);
                b0 = *(const __attribute__((address_space(1))) v4u*)(pb + (it & 63) * 3072);
                b1 = *(const __attribute__((address_space(1))) v4u*)(pb + (it & 63) * 3072 + 1024);
                b2 = *(const __attribute__((address_space(1))) v4u*)(pb + (it & 63) * 3072 + 2048);
            }
            if (MODE == 5) {                                   // pin the interleave: 1 MFMA : 2 VALU
#pragma unroll
                for (int r = 0; r < 24; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x2, 2, 0);
                }
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s + a0.x + (float)b0.x;
}


// MODE 6: pre-split operands, LDS-DMA staging.  KS = k-steps (of 16) per barrier.
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int BPC, int KS>
__global__ __launch_bounds__(256, BPC) void kglds(float* out, const unsigned* __restrict__ ga3, const unsigned* __restrict__ gb3, int iters)
{
    constexpr int OPD = 3 * 128 * 8;                                // dwords of one operand image of one k-step: [term 3][row 128][8]
    constexpr int BUF = KS * 2 * OPD;                               // [k-step][A | B]
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 2 * BUF; i += 256) lds[i] = 0x3f803f80u + (i & 7);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int g = lane >> 5, il = lane & 31, wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
    // this wave's three 1-KiB DMA pieces per operand and k-step: piece c = wid * 3 + s -> term c / 4, rows 32 (c % 4) .. + 31;
    // lane l brings row + (l >> 1), 16-byte half (l & 1) ^ ((row >> 3) & 1)
    unsigned srcoff[3], dstoff[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int c = wid * 3 + s, term = c >> 2, row = (c & 3) * 32 + (lane >> 1);
        const int half = (lane & 1) ^ ((row >> 3) & 1);
        srcoff[s] = (unsigned)row * (64 * 24) + term * 8 + half * 4;          // dwords; [row][k-step 64][24]
        dstoff[s] = (term * 128 + (c & 3) * 32) * 8;                           // dwords, wave-uniform (+ lane * 4 by the hardware)
    }
    const unsigned* pa = ga3 + (size_t)(blockIdx.x & 63) * 128 * 64 * 24;
    const unsigned* pb = gb3;
    auto dma = [&](int it, unsigned* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = ((it * KS + ks) & 63) * 24;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                __builtin_amdgcn_global_load_lds((gptr_t)(pa + srcoff[s] + kk), (lptr_t)(buf + ks * 2 * OPD + dstoff[s]), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr_t)(pb + srcoff[s] + kk), (lptr_t)(buf + ks * 2 * OPD + OPD + dstoff[s]), 16, 0, 0);
            }
        }
    };
    // fragment addresses: row's 16-byte half g sits in slot g ^ ((row >> 3) & 1)
    int fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm0 + i * 32 + il, rb = wn0 + i * 32 + il;
        fa[i] = ra * 8 + ((g ^ ((ra >> 3) & 1)) << 2);
        fb[i] = rb * 8 + ((g ^ ((rb >> 3) & 1)) << 2);
    }
    dma(0, lds);
    __syncthreads();
    int cur = 0;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const unsigned* cb = lds + cur * BUF;
        dma(it + 1, lds + (cur ^ 1) * BUF);                          // next k-step(s): in flight under this step's MFMAs
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned* As = cb + ks * 2 * OPD;
            const unsigned* Bs = As + OPD;
            v4u af[2][3], bf[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    af[i][t] = *reinterpret_cast<const v4u*>(As + t * 128 * 8 + fa[i]);
                    bf[i][t] = *reinterpret_cast<const v4u*>(Bs + t * 128 * 8 + fb[i]);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = mm(af[i][2], bf[j][0], acc[i][j]);
                    acc[i][j] = mm(af[i][1], bf[j][1], acc[i][j]);
                    acc[i][j] = mm(af[i][0], bf[j][2], acc[i][j]);
                    acc[i][j] = mm(af[i][1], bf[j][0], acc[i][j]);
                    acc[i][j] = mm(af[i][0], bf[j][1], acc[i][j]);
                    acc[i][j] = mm(af[i][0], bf[j][0], acc[i][j]);
                }
        }
        __syncthreads();                                             // (vmcnt(0) + barrier: the DMAs have landed, everyone is done reading `cur`)
        cur ^= 1;
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

// MODE 6b: as kglds, but the DMA runs TWO k-steps ahead of the MFMAs through THREE LDS buffers, retired with a COUNTED s_waitcnt
// vmcnt(6) (= this thread's 6 DMAs of the newest k-step may stay in flight) and a RAW s_barrier -- __syncthreads() would emit vmcnt(0)
// and drain the newest DMA at every k-step (cdna_hip_programming.md "Pipelining across barriers").  NB = LDS buffers (3 or 4).
// LAY: global layout of the pre-split operands -- 0: [row][k-step][term][16 bf16] (one 96-byte record per row and k-step: a DMA
// instruction touches 32 records, 32 contiguous bytes each); 1: [k-step][term][row][16 bf16] (planes in the consumer's k-step order: a
// DMA instruction reads 1 KiB contiguous)
template <int BPC, int NB, int LAY>
__global__ __launch_bounds__(256, BPC) void kglds3(float* out, const unsigned* __restrict__ ga3, const unsigned* __restrict__ gb3, int iters)
{
    constexpr int OPD = 3 * 128 * 8, BUF = 2 * OPD;
    __shared__ __attribute__((aligned(16))) unsigned lds[NB * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < NB * BUF; i += 256) lds[i] = 0x3f803f80u + (i & 7);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int g = lane >> 5, il = lane & 31, wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
    unsigned srcA[3], srcB[3], dstoff[3];
    constexpr unsigned ROWS_A = 64 * 128, ROWS_B = 128;              // rows of the whole A / B matrices
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int c = wid * 3 + s, term = c >> 2, row = (c & 3) * 32 + (lane >> 1);
        const int half = (lane & 1) ^ ((row >> 3) & 1);
        if (LAY == 0) srcA[s] = srcB[s] = (unsigned)row * (64 * 24) + term * 8 + half * 4;
        else { srcA[s] = ((unsigned)term * ROWS_A + row) * 8 + half * 4; srcB[s] = ((unsigned)term * ROWS_B + row) * 8 + half * 4; }
        dstoff[s] = (term * 128 + (c & 3) * 32) * 8;
    }
    const unsigned* pa = ga3 + (LAY == 0 ? (size_t)(blockIdx.x & 63) * 128 * 64 * 24 : (size_t)(blockIdx.x & 63) * 128 * 8);
    const unsigned* pb = gb3;
    auto dma = [&](int it, unsigned* buf) __attribute__((always_inline)) {
        const unsigned ka = LAY == 0 ? (it & 63) * 24 : (unsigned)(it & 63) * (3 * ROWS_A * 8);
        const unsigned kb = LAY == 0 ? (it & 63) * 24 : (unsigned)(it & 63) * (3 * ROWS_B * 8);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            __builtin_amdgcn_global_load_lds((gptr_t)(pa + srcA[s] + ka), (lptr_t)(buf + dstoff[s]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(pb + srcB[s] + kb), (lptr_t)(buf + OPD + dstoff[s]), 16, 0, 0);
        }
    };
    int fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm0 + i * 32 + il, rb = wn0 + i * 32 + il;
        fa[i] = ra * 8 + ((g ^ ((ra >> 3) & 1)) << 2);
        fb[i] = rb * 8 + ((g ^ ((rb >> 3) & 1)) << 2);
    }
#pragma unroll
    for (int d = 0; d < NB - 1; ++d) dma(d, lds + d * BUF);
    if (NB == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0, nxt = NB - 1;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const unsigned* As = lds + cur * BUF;
        const unsigned* Bs = As + OPD;
        dma(it + NB - 1, lds + nxt * BUF);                           // NB - 1 k-steps ahead: its buffer was read in iteration it - 1
        v4u af[2][3], bf[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                af[i][t] = *reinterpret_cast<const v4u*>(As + t * 128 * 8 + fa[i]);
                bf[i][t] = *reinterpret_cast<const v4u*>(Bs + t * 128 * 8 + fb[i]);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = mm(af[i][2], bf[j][0], acc[i][j]);
                acc[i][j] = mm(af[i][1], bf[j][1], acc[i][j]);
                acc[i][j] = mm(af[i][0], bf[j][2], acc[i][j]);
                acc[i][j] = mm(af[i][1], bf[j][0], acc[i][j]);
                acc[i][j] = mm(af[i][0], bf[j][1], acc[i][j]);
                acc[i][j] = mm(af[i][0], bf[j][0], acc[i][j]);
            }
        // the k-step read next must have landed (all but the newest NB - 2 k-steps' DMAs of this thread), fragment reads of `cur` retired
        if (NB == 3) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur + 1 == NB ? 0 : cur + 1;
        nxt = nxt + 1 == NB ? 0 : nxt + 1;
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

// MODE 6c: the pipelined plane-layout loop on a 256 x 128 block tile (8 waves, each 64 x 64 as before): 36 KB of DMA per k-step for
// twice the MFMAs of the 128 x 128 tile (-25 % operand bytes per flop).  3 LDS buffers = 108 KB: one block per CU, two waves per SIMD.
__global__ __launch_bounds__(512, 1) void kglds256(float* out, const unsigned* __restrict__ ga3, const unsigned* __restrict__ gb3, int iters)
{
    constexpr int NB = 3, OPA = 3 * 256 * 8, OPB = 3 * 128 * 8, BUF = OPA + OPB;
    __shared__ __attribute__((aligned(16))) unsigned lds[NB * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < NB * BUF; i += 512) lds[i] = 0x3f803f80u + (i & 7);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int g = lane >> 5, il = lane & 31, wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
    constexpr unsigned ROWS_A = 64 * 256, ROWS_B = 128;
    // 36 one-KiB pieces per k-step: 24 of A (term c / 8, rows 32 (c % 8) ..), 12 of B; wave w brings pieces w, w + 8, ..., < 36
    unsigned src[5], dst[5]; bool isB[5];
    const int npc = wid < 4 ? 5 : 4;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int c = wid + 8 * s;
        const bool b = c >= 24;
        const int cc = b ? c - 24 : c, rb = b ? 4 : 8;
        const int term = cc / rb, row = (cc % rb) * 32 + (lane >> 1), half = (lane & 1) ^ ((row >> 3) & 1);
        src[s] = ((unsigned)term * (b ? ROWS_B : ROWS_A) + row) * 8 + half * 4;
        dst[s] = (b ? OPA : 0) + (term * (b ? 128 : 256) + (cc % rb) * 32) * 8;
        isB[s] = b;
    }
    const unsigned* pa = ga3 + (size_t)(blockIdx.x & 63) * 256 * 8;
    auto dma = [&](int it, unsigned* buf) __attribute__((always_inline)) {
        const unsigned ka = (unsigned)(it & 31) * (3 * ROWS_A * 8), kb = (unsigned)(it & 31) * (3 * ROWS_B * 8);
#pragma unroll
        for (int s = 0; s < 5; ++s)
            if (s < 4 || npc == 5)
                __builtin_amdgcn_global_load_lds((gptr_t)(isB[s] ? gb3 + src[s] + kb : pa + src[s] + ka), (lptr_t)(buf + dst[s]), 16, 0, 0);
    };
    int fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = wm0 + i * 32 + il, rb = wn0 + i * 32 + il;
        fa[i] = ra * 8 + ((g ^ ((ra >> 3) & 1)) << 2);
        fb[i] = rb * 8 + ((g ^ ((rb >> 3) & 1)) << 2);
    }
    dma(0, lds); dma(1, lds + BUF);
    if (npc == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0, nxt = 2;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const unsigned* As = lds + cur * BUF;
        const unsigned* Bs = As + OPA;
        dma(it + 2, lds + nxt * BUF);
        v4u af[2][3], bf[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                af[i][t] = *reinterpret_cast<const v4u*>(As + t * 256 * 8 + fa[i]);
                bf[i][t] = *reinterpret_cast<const v4u*>(Bs + t * 128 * 8 + fb[i]);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = mm(af[i][2], bf[j][0], acc[i][j]);
                acc[i][j] = mm(af[i][1], bf[j][1], acc[i][j]);
                acc[i][j] = mm(af[i][0], bf[j][2], acc[i][j]);
                acc[i][j] = mm(af[i][1], bf[j][0], acc[i][j]);
                acc[i][j] = mm(af[i][0], bf[j][1], acc[i][j]);
                acc[i][j] = mm(af[i][0], bf[j][0], acc[i][j]);
            }
        if (npc == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur = cur == 2 ? 0 : cur + 1;
        nxt = nxt == 2 ? 0 : nxt + 1;
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 512 + tid] = s;
}

// the f32 MFMA loop (as tools/micro/mfma_peak.hip MODE 1, 2 x 2 tiles per wave: 4 A/B b128 reads per 16 MFMAs of 32x32x2)
template <int BPC>
__global__ __launch_bounds__(256, BPC) void kf32(float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * 256 * 20];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int i = tid; i < 2 * 256 * 20; i += 256) lds[i] = 0.001f * (i & 63);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int g = lane >> 5, il = lane & 31, wm0 = (wid >> 1) * 64, wn0 = (wid & 1) * 64;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const float* As = lds + (it & 1) * 256 * 20;
        const float* Bs = As + 128 * 20;
        v4f af[2][2], bf[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[h][i] = *reinterpret_cast<const v4f*>(As + (wm0 + i * 32 + il) * 20 + h * 8 + g * 4);
                bf[h][i] = *reinterpret_cast<const v4f*>(Bs + (wn0 + i * 32 + il) * 20 + h * 8 + g * 4);
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[h][i][e], bf[h][j][e], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

static float* g_out; static float* g_a; static unsigned* g_b; static unsigned* g_a3; static unsigned* g_b3;

template <int MODE, int BPC> double run(const char* name)
{
    const int iters = 4000, grid = 256 * BPC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, BPC><<<grid, 256>>>(g_out, g_a, g_b, 50);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE, BPC><<<grid, 256>>>(g_out, g_a, g_b, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double eq = (double)grid * 4 * iters * 4 * 2.0 * 32 * 32 * 16;      // fp32-equivalent flops
    const double tf = eq / best / 1e9;
    printf("%-66s blocks/CU=%d  %8.3f ms  %7.1f TF fp32-equivalent  (%6.1f TF bf16 MFMA issued)\n", name, BPC, best, tf, tf * 6);
    return tf;
}

template <int BPC, int KS> double runglds(const char* name)
{
    const int iters = 4000 / KS, grid = 256 * BPC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kglds<BPC, KS><<<grid, 256>>>(g_out, g_a3, g_b3, 50);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kglds<BPC, KS><<<grid, 256>>>(g_out, g_a3, g_b3, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double tf = (double)grid * 4 * iters * KS * 4 * 2.0 * 32 * 32 * 16 / best / 1e9;
    printf("%-66s blocks/CU=%d  %8.3f ms  %7.1f TF fp32-equivalent  (%6.1f TF bf16 MFMA issued)\n", name, BPC, best, tf, tf * 6);
    return tf;
}

template <int BPC, int NB, int LAY> double runglds3(const char* name)
{
    const int iters = 4000, grid = 256 * BPC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kglds3<BPC, NB, LAY><<<grid, 256>>>(g_out, g_a3, g_b3, 50);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kglds3<BPC, NB, LAY><<<grid, 256>>>(g_out, g_a3, g_b3, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double tf = (double)grid * 4 * iters * 4 * 2.0 * 32 * 32 * 16 / best / 1e9;
    printf("%-66s blocks/CU=%d  %8.3f ms  %7.1f TF fp32-equivalent  (%6.1f TF bf16 MFMA issued)\n", name, BPC, best, tf, tf * 6);
    return tf;
}

double runglds256(const char* name)
{
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kglds256<<<grid, 512>>>(g_out, g_a3, g_b3, 50);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kglds256<<<grid, 512>>>(g_out, g_a3, g_b3, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double tf = (double)grid * 8 * iters * 4 * 2.0 * 32 * 32 * 16 / best / 1e9;
    printf("%-66s blocks/CU=%d  %8.3f ms  %7.1f TF fp32-equivalent  (%6.1f TF bf16 MFMA issued)\n", name, 1, best, tf, tf * 6);
    return tf;
}
template <int BPC> double runf32()
{
    const int iters = 4000, grid = 256 * BPC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kf32<BPC><<<grid, 256>>>(g_out, 50);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kf32<BPC><<<grid, 256>>>(g_out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double tf = (double)grid * 4 * iters * 4 * 2.0 * 32 * 32 * 16 / best / 1e9;
    printf("%-66s blocks/CU=%d  %8.3f ms  %7.1f TF fp32\n", "f32 MFMA loop (32x32x2), LDS-fed 2x2 tiles, no staging", BPC, best, tf);
    return tf;
}
int main()
{
    hipMalloc(&g_out, 256 * 4 * 512 * sizeof(float));
    hipMalloc(&g_a, 64 * 128 * 1024 * sizeof(float)); hipMemset(g_a, 0, 64 * 128 * 1024 * sizeof(float));
    hipMalloc(&g_b, 64 * 3072 * sizeof(unsigned)); hipMemset(g_b, 0, 64 * 3072 * sizeof(unsigned));
    run<0, 1>("bf16x3 register-fed (24 MFMA / k-step)");
    run<0, 2>("bf16x3 register-fed (24 MFMA / k-step)");
    run<1, 1>("bf16x3 + 12 ds_read_b128");
    run<1, 2>("bf16x3 + 12 ds_read_b128");
    run<2, 1>("bf16x3 + reads + split VALU (8 fp32 / thread) + 6 ds_write_b128 + barrier");
    const double s2 = run<2, 2>("bf16x3 + reads + split VALU (8 fp32 / thread) + 6 ds_write_b128 + barrier");
    run<3, 1>("bf16x3 + reads + split + writes + global loads (A fp32, B pre-split)");
    const double s3 = run<3, 2>("bf16x3 + reads + split + writes + global loads (A fp32, B pre-split)");
    run<4, 1>("bf16x3 pipelined: split interleaved with the MFMAs, loads 2 k-steps ahead");
    const double s4 = run<4, 2>("bf16x3 pipelined: split interleaved with the MFMAs, loads 2 k-steps ahead");
    run<5, 1>("bf16x3 pipelined + sched_group_barrier (1 MFMA : 2 VALU)");
    run<5, 2>("bf16x3 pipelined + sched_group_barrier (1 MFMA : 2 VALU)");
    hipMalloc(&g_a3, (size_t)64 * 128 * 64 * 24 * 4); hipMemset(g_a3, 0x3f, (size_t)64 * 128 * 64 * 24 * 4);      // 64 row blocks, pre-split
    hipMalloc(&g_b3, (size_t)128 * 64 * 24 * 4); hipMemset(g_b3, 0x3f, (size_t)128 * 64 * 24 * 4);
    runglds<1, 1>("bf16x3 stage C: both operands pre-split, LDS-DMA (K = 16 / barrier)");
    const double c1 = runglds<2, 1>("bf16x3 stage C: both operands pre-split, LDS-DMA (K = 16 / barrier)");
    const double c2 = runglds<1, 2>("bf16x3 stage C: both operands pre-split, LDS-DMA (K = 32 / barrier)");
    runglds3<1, 3, 0>("bf16x3 stage C pipelined: DMA 2 k-steps ahead, 3 LDS buffers, vmcnt(6) + raw s_barrier");
    const double c3 = runglds3<2, 3, 0>("bf16x3 stage C pipelined: DMA 2 k-steps ahead, 3 LDS buffers, vmcnt(6) + raw s_barrier");
    const double c4 = runglds3<1, 4, 0>("bf16x3 stage C pipelined: DMA 3 k-steps ahead, 4 LDS buffers, vmcnt(12) + raw s_barrier");
    runglds3<1, 3, 1>("bf16x3 stage C pipelined, PLANE layout [k-step][term][row][16] (1 KiB per DMA), 3 buffers");
    const double c5 = runglds3<2, 3, 1>("bf16x3 stage C pipelined, PLANE layout [k-step][term][row][16] (1 KiB per DMA), 3 buffers");
    const double c6 = runglds3<1, 4, 1>("bf16x3 stage C pipelined, PLANE layout, 4 buffers");
    const double c7 = runglds256("bf16x3 stage C pipelined, PLANE layout, 256 x 128 block tile (8 waves), 3 buffers");
    double best = c1 > c2 ? c1 : c2; best = c3 > best ? c3 : best; best = c4 > best ? c4 : best; best = c5 > best ? c5 : best; best = c6 > best ? c6 : best;
    best = c7 > best ? c7 : best;
    printf("plane layout: %.1f (3 buffers, 2 blocks/CU), %.1f (4 buffers, 1 block/CU), %.1f (256 x 128 tile, 8 waves)\n", c5, c6, c7);
    printf("stage C go / no-go (>= 330 TF fp32-equivalent in the loop): %.1f (K16, 2 blocks/CU), %.1f (K32, 1 block/CU), %.1f (3 buffers, 2 blocks/CU), "
           "%.1f (4 buffers, 1 block/CU) -> %s\n", c1, c2, c3, c4, best >= 330.0 ? "GO" : "NO-GO");
    runf32<1>();
    const double f = runf32<2>();
    printf("ratio: staged bf16x3 loop / f32 MFMA loop = %.2f (LDS-staged), %.2f (with global loads), %.2f (pipelined)   [kill criterion: < 1.3]\n", s2 / f, s3 / f, s4 / f);
    return 0;
}
