// fp32 contraction through the bf16 matrix pipe: is it faster than the fp32 MFMA, and as exact?
//
// An fp32 value is the exact sum of three bf16 pieces (8 + 8 + 8 significand bits): x = x1 + x2 + x3.  A product a * b
// then is the sum of nine exact bf16 x bf16 products, of which the six with i + j <= 4 carry everything above 2^-24 of
// |a * b| — the size of fp32's own rounding.  v_mfma_f32_32x32x16_bf16 runs 16 x the MACs per cycle of
// v_mfma_f32_32x32x2_f32, so six of them cost 6 / 16 of the fp32 instruction's time; the price is the split itself
// (VALU, which on gfx950 does not overlap with MFMA — tools/mfma_valu_cost.hip).
//
// The loop here has the shape of one tap step of the weight-stationary conv GEMM (csrc/conv_gemm.hip: wstat_body): the
// lane's operand row arrives as 32 fp32 values (8 x 16-byte loads, L2-resident here), the 64 x 64 weight tile sits in
// LDS and its fragments are read per MFMA group.
//   MODE 0: fp32 MFMA, 32-row wave tile (64 MFMAs per step)            — what ships
//   MODE 1: split in registers + bf16 MFMA, 32-row wave tile (48 MFMAs per step)
//   MODE 2: the same with a 64-row wave tile (two operand rows per lane, 96 MFMAs per step, weight fragments shared)
//   MODE 4: MODE 1 with v_mfma_f32_16x16x32_bf16 (2 x 4 tiles of 16 x 16 per wave, 96 MFMAs per step): the shape the chip
//           clocks higher on
//   MODE 3: MODE 1 software-pipelined: the split of step s + 1 is interleaved with the MFMAs of step s (three VALU per
//           MFMA gap, __builtin_amdgcn_sched_group_barrier), the loads of step s + 2 follow the split that freed
//           their registers — a bf16 MFMA holds the vector issue for 8 of its 32 cycles only, the rest can hide VALU
// Prints shader cycles per step per wave and the error of the result against float64 on the host.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LDA = 68;        // fp32 weight rows in LDS (floats)
constexpr int LDB = 72;        // bf16 weight rows in LDS (bf16 elements: 144 bytes, 16-byte aligned, odd multiple of 16 B)

struct Split3 { unsigned p1, p2, p3; };       // two values per register: low half = first value's piece

// round-to-nearest pieces: x = x1 + x2 + x3 exactly (each residual is exact in fp32)
// Selectors of v_dot2c_f32_bf16 (a.lo b.lo + a.hi b.hi + c) that subtract one half of a packed pair: (-1, 0) and
// (0, -1).  Made once per kernel and kept in registers: as an inline constant the assembler's -1.0 is not the
// packed-bf16 pattern.
struct SplitSel { bf16x2 m10, m01; };
__device__ __forceinline__ SplitSel make_split_sel() {
    unsigned u10 = 0x0000bf80u, u01 = 0xbf800000u;
    asm volatile("" : "+v"(u10), "+v"(u01));
    return SplitSel{__builtin_bit_cast(bf16x2, u10), __builtin_bit_cast(bf16x2, u01)};
}
__device__ __forceinline__ Split3 split_pair(float x, float y, const SplitSel& q) {
    // the residual x - bf16(x) in ONE instruction per value (exact: the products are exact and the difference is
    // representable)
    Split3 s;
    const bf16x2 h1 = __builtin_convertvector(floatx2{x, y}, bf16x2);
    const float rx = __builtin_amdgcn_fdot2_f32_bf16(h1, q.m10, x, false), ry = __builtin_amdgcn_fdot2_f32_bf16(h1, q.m01, y, false);
    const bf16x2 h2 = __builtin_convertvector(floatx2{rx, ry}, bf16x2);
    const float sx = __builtin_amdgcn_fdot2_f32_bf16(h2, q.m10, rx, false), sy = __builtin_amdgcn_fdot2_f32_bf16(h2, q.m01, ry, false);
    const bf16x2 h3 = __builtin_convertvector(floatx2{sx, sy}, bf16x2);
    s.p1 = __builtin_bit_cast(unsigned, h1);
    s.p2 = __builtin_bit_cast(unsigned, h2);
    s.p3 = __builtin_bit_cast(unsigned, h3);
    return s;
}

struct Frag3 { uint4 p[3]; };
__device__ __forceinline__ Frag3 split8(const float4& u, const float4& v, const SplitSel& q) {
    const Split3 a = split_pair(u.x, u.y, q), b = split_pair(u.z, u.w, q), c = split_pair(v.x, v.y, q), d = split_pair(v.z, v.w, q);
    Frag3 f;
    f.p[0] = make_uint4(a.p1, b.p1, c.p1, d.p1);
    f.p[1] = make_uint4(a.p2, b.p2, c.p2, d.p2);
    f.p[2] = make_uint4(a.p3, b.p3, c.p3, d.p3);
    return f;
}
__device__ __forceinline__ bf16x8 as_bf(const uint4& u) { return __builtin_bit_cast(bf16x8, u); }

__device__ __forceinline__ floatx16 mfma6(const Frag3& a, const Frag3& b, floatx16 c) {
    // small terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[2]), as_bf(b.p[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[0]), as_bf(b.p[2]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[1]), as_bf(b.p[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[1]), as_bf(b.p[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[0]), as_bf(b.p[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[0]), as_bf(b.p[0]), c, 0, 0, 0);
    return c;
}

typedef float floatx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ floatx4 mfma6_16(const Frag3& a, const Frag3& b, floatx4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[2]), as_bf(b.p[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[0]), as_bf(b.p[2]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[1]), as_bf(b.p[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[1]), as_bf(b.p[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[0]), as_bf(b.p[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[0]), as_bf(b.p[0]), c, 0, 0, 0);
    return c;
}

// A: [rows][64] fp32, W: [64 co][64 ci] fp32, out: [rows][64].  Every wave works on its own rows; `steps` tap steps
// re-read the same rows (the timing build) or steps == 1 (the check).
template <int MODE>
__global__ __launch_bounds__(256, MODE == 2 ? 1 : 2) void k(const float* __restrict__ A, const float* __restrict__ W,
                                                             float* __restrict__ out, int steps, int a_mask,
                                                             long long* cycles) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, h = lane >> 5;
    const SplitSel ssel = make_split_sel();
    constexpr int RW = MODE == 2 ? 64 : 32;                         // rows per wave
    if (MODE == 0) {
        float* Ws = reinterpret_cast<float*>(smem);
        for (int e = tid; e < 64 * 64; e += 256) Ws[(e >> 6) * LDA + (e & 63)] = W[e];
    } else {
        // three planes [64 co][LDB]; channel c = 8 kk + 4 h' + j sits at 16 (kk / 2) + 8 h' + 4 (kk & 1) + j, so that
        // the 8 channels of one MFMA step and lane half are 16 contiguous bytes
        __bf16* Wp = reinterpret_cast<__bf16*>(smem);
        for (int e = tid; e < 64 * 32; e += 256) {
            const int co = e >> 5, c = 2 * (e & 31);
            const Split3 s = split_pair(W[co * 64 + c], W[co * 64 + c + 1], ssel);
            const int kk = c >> 3, hh = (c >> 2) & 1, j = c & 3;
            const int pos = MODE == 4 ? c : 16 * (kk >> 1) + 8 * hh + 4 * (kk & 1) + j;      // MODE 4: natural order
            *reinterpret_cast<unsigned*>(&Wp[(0 * 64 + co) * LDB + pos]) = s.p1;
            *reinterpret_cast<unsigned*>(&Wp[(1 * 64 + co) * LDB + pos]) = s.p2;
            *reinterpret_cast<unsigned*>(&Wp[(2 * 64 + co) * LDB + pos]) = s.p3;
        }
    }
    __syncthreads();
    const long row0 = ((long)blockIdx.x * 4 + w) * RW;
    floatx16 acc[MODE == 2 ? 4 : 2] = {};
    float4 ra[MODE == 2 ? 16 : 8];
    auto load = [&](int s) {
        const float* p = A + (((row0 + i) * 64 + (long)s * 8192) & a_mask) + 4 * h;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ra[kk] = *reinterpret_cast<const float4*>(p + 8 * kk);
        if (MODE == 2) {
            const float* p2 = A + (((row0 + 32 + i) * 64 + (long)s * 8192) & a_mask) + 4 * h;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) ra[8 + kk] = *reinterpret_cast<const float4*>(p2 + 8 * kk);
        }
    };
    load(0);
    const long long t0 = clock64();
    if (MODE == 4) {
        const int m = lane & 15, kg = lane >> 4;
        floatx4 c4[2][4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) c4[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
        float4 r4[8];
        auto load4 = [&](int s) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const float* p = A + (((row0 + 16 * mi + m) * 64 + (long)s * 8192) & a_mask) + 32 * ks + 8 * kg;
                    r4[(mi * 2 + ks) * 2] = *reinterpret_cast<const float4*>(p);
                    r4[(mi * 2 + ks) * 2 + 1] = *reinterpret_cast<const float4*>(p + 4);
                }
        };
        load4(0);
        const __bf16* Wp = reinterpret_cast<const __bf16*>(smem);
        for (int s = 0; s < steps; ++s) {
            Frag3 fa[2][2];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fa[mi][ks] = split8(r4[(mi * 2 + ks) * 2], r4[(mi * 2 + ks) * 2 + 1], ssel);
            load4(s + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    Frag3 b;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        b.p[pl] = *reinterpret_cast<const uint4*>(&Wp[(pl * 64 + 16 * ni + m) * LDB + 32 * ks + 8 * kg]);
                    c4[0][ni] = mfma6_16(fa[0][ks], b, c4[0][ni]);
                    c4[1][ni] = mfma6_16(fa[1][ks], b, c4[1][ni]);
                }
        }
        const long long t1 = clock64();
        if (lane == 0 && cycles) cycles[blockIdx.x * 4 + w] = t1 - t0;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(row0 + 16 * mi + 4 * kg + r) * 64 + 16 * ni + m] = c4[mi][ni][r];
        return;
    }
    if (MODE == 3) {
        Frag3 fa[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[ks] = split8(ra[2 * ks], ra[2 * ks + 1], ssel);
        load(1);
        const __bf16* Wp = reinterpret_cast<const __bf16*>(smem);
        for (int s = 0; s < steps; ++s) {
            Frag3 fn[4];
            const float* pn = A + (((row0 + i) * 64 + (long)(s + 2) * 8192) & a_mask) + 4 * h;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                Frag3 b0, b1;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    b0.p[pl] = *reinterpret_cast<const uint4*>(&Wp[(pl * 64 + i) * LDB + 16 * ks + 8 * h]);
                    b1.p[pl] = *reinterpret_cast<const uint4*>(&Wp[(pl * 64 + 32 + i) * LDB + 16 * ks + 8 * h]);
                }
                fn[ks] = split8(ra[2 * ks], ra[2 * ks + 1], ssel);                   // step s + 1
                ra[2 * ks] = *reinterpret_cast<const float4*>(pn + 16 * ks);     // step s + 2
                ra[2 * ks + 1] = *reinterpret_cast<const float4*>(pn + 16 * ks + 8);
                acc[0] = mfma6(fa[ks], b0, acc[0]);
                acc[1] = mfma6(fa[ks], b1, acc[1]);
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);               // the six fragment reads
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);           // three VALU of the split
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);               // the two loads
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[ks] = fn[ks];
        }
    }
    for (int s = 0; MODE != 3 && s < steps; ++s) {
        if (MODE == 0) {
            float4 a[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) a[kk] = ra[kk];
            load(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            const float* Ws = reinterpret_cast<const float*>(smem);
            const float* bp0 = &Ws[i * LDA + 4 * h];
            const float* bp1 = bp0 + 32 * LDA;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float4 p4 = *reinterpret_cast<const float4*>(bp0 + 8 * kk);
                const float4 r4 = *reinterpret_cast<const float4*>(bp1 + 8 * kk);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].x, p4.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].x, r4.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].y, p4.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].y, r4.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].z, p4.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].z, r4.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].w, p4.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk].w, r4.w, acc[1], 0, 0, 0);
            }
        } else {
            constexpr int NR = MODE == 2 ? 2 : 1;
            Frag3 fa[NR][4];
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[r][ks] = split8(ra[8 * r + 2 * ks], ra[8 * r + 2 * ks + 1], ssel);
            load(s + 1);
            __builtin_amdgcn_sched_barrier(0);
            const __bf16* Wp = reinterpret_cast<const __bf16*>(smem);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                Frag3 b0, b1;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    b0.p[pl] = *reinterpret_cast<const uint4*>(&Wp[(pl * 64 + i) * LDB + 16 * ks + 8 * h]);
                    b1.p[pl] = *reinterpret_cast<const uint4*>(&Wp[(pl * 64 + 32 + i) * LDB + 16 * ks + 8 * h]);
                }
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    acc[2 * r] = mfma6(fa[r][ks], b0, acc[2 * r]);
                    acc[2 * r + 1] = mfma6(fa[r][ks], b1, acc[2 * r + 1]);
                }
            }
        }
    }
    const long long t1 = clock64();
    if (lane == 0 && cycles) cycles[blockIdx.x * 4 + w] = t1 - t0;
    // rows of the accumulator registers: (reg & 3) + 8 (reg >> 2) + 4 h; column i (+ 32)
#pragma unroll
    for (int r = 0; r < (MODE == 2 ? 2 : 1); ++r)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const long row = row0 + 32 * r + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            out[row * 64 + i] = acc[2 * r][reg];
            out[row * 64 + 32 + i] = acc[2 * r + 1][reg];
        }
}

static int g_mask_bits = 21;
template <int MODE>
static void run(const float* dA, const float* dW, float* dO, long long* dC, const std::vector<float>& hA,
                const std::vector<float>& hW, int rows_chk) {
    const size_t lds = MODE == 0 ? 64 * LDA * 4 : 3 * 64 * LDB * 2;
    CHECK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    constexpr int RW = MODE == 2 ? 64 : 32;
    // check: one step, a few workgroups
    const int blocks_chk = rows_chk / (4 * RW);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks_chk), dim3(256), lds, 0, dA, dW, dO, 1, 0x7fffffff, nullptr);
    CHECK(hipDeviceSynchronize());
    std::vector<float> hO((size_t)rows_chk * 64);
    CHECK(hipMemcpy(hO.data(), dO, hO.size() * 4, hipMemcpyDeviceToHost));
    double max_err = 0, sum_err = 0, max_ref = 0, bias = 0;
    for (int r = 0; r < rows_chk; ++r)
        for (int co = 0; co < 64; ++co) {
            double ref = 0, mag = 0;
            for (int c = 0; c < 64; ++c) { ref += (double)hA[(size_t)r * 64 + c] * hW[co * 64 + c]; mag += fabs((double)hA[(size_t)r * 64 + c] * hW[co * 64 + c]); }
            const double e = (hO[(size_t)r * 64 + co] - ref) / mag;          // relative to sum |a b|
            max_err = fmax(max_err, fabs(e)); sum_err += fabs(e); bias += e; max_ref = fmax(max_ref, fabs(ref));
        }
    const double n = (double)rows_chk * 64;
    // timing: every CU busy (MODE 2: one workgroup of 4 waves x 64 rows per CU; else two of 4 x 32), operand from L2
    const int steps = 2000;
    const int blocks = MODE == 2 ? 256 : 512;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), lds, 0, dA, dW, dO, steps, (1 << g_mask_bits) - 1, dC);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<long long> hC(blocks * 4);
    CHECK(hipMemcpy(hC.data(), dC, hC.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0;
    for (auto c : hC) cyc += (double)c;
    cyc /= hC.size() * (double)steps;
    const double flop = (double)blocks * 4 * RW * 64 * 64 * 2 * steps;
    printf("MODE %d: %8.1f clock64 ticks per step per wave, %7.3f ms, %6.1f TFLOP/s (fp32-equivalent) | error / sum|ab|: max %.2e mean %.2e bias %+.2e (units of 2^-24: max %.2f mean %.3f)\n",
           MODE, cyc, ms, flop / ms * 1e-9, max_err, sum_err / n, bias / n, max_err * 16777216.0, sum_err / n * 16777216.0);
}

int main(int argc, char** argv) {
    if (argc > 1) g_mask_bits = atoi(argv[1]);       // operand footprint of the timing runs: 2^bits floats
    printf("operand footprint %d KB\n", (1 << g_mask_bits) / 256);
    const int rows = 1 << 15;                        // 8 MB of operand
    std::vector<float> hA((size_t)rows * 64), hW(64 * 64);
    srand(3);
    auto rnd = [] { return (float)((rand() / (double)RAND_MAX * 2 - 1) * exp((rand() / (double)RAND_MAX - 0.5) * 6)); };
    for (auto& v : hA) v = rnd();
    for (auto& v : hW) v = rnd();
    float *dA, *dW, *dO; long long* dC;
    CHECK(hipMalloc(&dA, hA.size() * 4)); CHECK(hipMalloc(&dW, hW.size() * 4));
    CHECK(hipMalloc(&dO, (size_t)(1 << 17) * 64 * 4)); CHECK(hipMalloc(&dC, 4096 * 8));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    run<0>(dA, dW, dO, dC, hA, hW, 1024);
    run<1>(dA, dW, dO, dC, hA, hW, 1024);
    run<2>(dA, dW, dO, dC, hA, hW, 1024);
    run<3>(dA, dW, dO, dC, hA, hW, 1024);
    run<4>(dA, dW, dO, dC, hA, hW, 1024);
    return 0;
}
