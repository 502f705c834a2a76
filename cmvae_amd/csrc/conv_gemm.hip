// fp32-MFMA gather-GEMM family for the dense contractions of the CMVAE hot path (gfx950).
//
// Every activation lives in HBM as a row-major matrix [pixels, C] (channels-last, C = 64 for
// the conv stacks => one pixel = one 256-B row).  One kernel shape serves conv 3x3, the four
// sub-pixel phases of a stride-2 transposed conv, their data gradients, and Linear layers:
//
//     out[r, 0:64] = act( sum_t  T(A[src(r,t), a_col_t : a_col_t+64]) @ W_t[64x64]  + bias )
//
// where r enumerates (image, y, x) over a per-image grid, src(r,t) = (image, y*sy+dy_t,
// x*sx+dx_t) (zero outside the image) and T is an optional per-channel affine + ReLU (the
// previous layer's BatchNorm folded into the operand load).  v_mfma_f32_32x32x2_f32 does the
// math (exact fp32 fma chains); the big launches (BF3 instantiations, conv_wgrad_bf3_multi_kernel) run the same fp32
// products as six bf16 MFMA products of exact three-piece operand splits (mfma_split6).  Kernels, by where the operands live:
//   conv_gemm_kernel               A and W_t tiles staged through LDS with register prefetch (folded launches > 9 taps)
//   conv_gemm_direct_kernel        A fragments straight from global memory, W_t tile through LDS per tap (2 barriers)
//   conv_gemm_wstat_kernel         all W_t of the launch resident in LDS, A from global memory, no barrier (<= 9 taps;
//                                  _multi_: up to four geometries side by side = the phases of a transposed conv;
//                                  ACCUM: a later pass over another tap range, adding to the output; BNB:
//                                  BatchNorm-backward sums epilogue; BF3: products on the bf16 pipe, <= 6 taps;
//                                  _splitk_: tap groups of one Linear data gradient side by side, partial products
//                                  added by splitk_reduce_kernel)
// The weight-gradient kernels contract over r instead (conv_wgrad_stream_kernel: fragments straight from global
// memory on v_mfma_f32_16x16x4_f32; conv_wgrad_bf3_multi_kernel: the same on the bf16 pipe for four-tap phases;
// conv_wgrad_kernel: LDS-staged, any shape).
#include "common.h"
#ifndef CMVAE_ABLATE
#define CMVAE_ABLATE 0
#endif
#ifndef CMVAE_STAGGER
#define CMVAE_STAGGER 0
#endif

namespace cmvae {
#if CMVAE_ABLATE == 30 || CMVAE_ABLATE == 73 || CMVAE_ABLATE == 74
// profiling builds: every weight-stationary workgroup logs (start, end) on the 100 MHz wall clock + tiles (30); every
// wave of the bf16-pipe weight gradient its waits for memory, shader cycles and loop time (73; 74: every load a hit)
__device__ unsigned long long g_wg_times[3 * 4096];
__device__ unsigned long long g_wg_quarters[4 * 4096];     // wall clock when the workgroup had done 1/4, 1/2, 3/4, 7/8 of its tiles
#endif

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;       // rows per block tile
constexpr int LDA = 68;       // LDS row pitch (floats): 16-B aligned, conflict-free b128 fragment reads

struct ConvGeom {
    int rows_total;
    int Hg, Wg;               // per-image grid the rows enumerate
    int Hi, Wi;               // input image, pixels
    int sy, sx;               // src = (y*sy + dy_t, x*sx + dx_t)
    int Ho, Wo;               // output image, pixels
    int oys, oxs, oy0, ox0;   // out pixel = (y*oys + oy0, x*oxs + ox0)
    int ntaps;
    int a_pitch, o_pitch;     // floats per input / output pixel row
    int w_pitch;              // floats between consecutive output columns of W_t ([col][k] layout)
    long w_nblk_stride;       // floats between 64-column output blocks (blockIdx.y)
    int dy[32], dx[32];       // dword arrays: indexed by the (uniform) tap counter they stay in SGPRs
    int a_col[32];
    int w_off[32];            // float offset of tap t's 64x64 weight block
    // pixel-major row order (0 = image-major): rows enumerate (y, x, image), r = (y*Wg + x)*pm_nimg + image.  A tile
    // of consecutive rows then holds ONE grid position of many images, so a tap that falls outside the image does so
    // for the whole tile and is skipped instead of multiplying zero padding (14 % of the taps of a 7x7 -> 14x14
    // transposed convolution, 31 % at 4x4 -> 7x7, 44 % at 2x2 -> 4x4).  Measured on the layer-3 launches (38400
    // images): forward phase 575 -> 512 us, weight gradient 635 -> 600 us; the 16-tap data gradient gains little
    // (2.34 -> 2.32 ms: its 16 gathered row sets per tile no longer overlap inside a tile, the saved MFMAs go into
    // memory traffic).  A class-major order (interior / edge / corner rectangles, image-major inside a class) was built
    // and measured as well: it keeps the locality but its edge tiles (1-2 taps per 32 KB of operand and output) are
    // bandwidth-bound and re-read the edge rows of the input: 545-570 / 650 / 2350 us, no better than image-major.
    int pm_nimg;
    int xcd_order;            // pixel-major launches: tiles dealt to the XCDs in image-group-major runs (wstat_body)
};

struct RowInfo {
    int off;                  // element offset of source pixel (image, y*sy, x*sx): pixel * a_pitch
    int ys, xs;               // y*sy, x*sx; a row past rows_total gets ys = INVALID_YS so that no tap is inside
};
constexpr int INVALID_YS = -(1 << 24);

// floor(a / d) for 0 <= a < 2^21 through the fp32 reciprocal of d (exact in that range; check_geom bounds Hg*Wg)
__device__ __forceinline__ int small_div(int a, float inv_d) { return (int)(((float)a + 0.5f) * inv_d); }

struct RowDecoder {
    float inv_hw, inv_wg;
    int hw, wg;
    __device__ __forceinline__ RowDecoder(const ConvGeom& g) : hw(g.Hg * g.Wg), wg(g.Wg) {
        inv_hw = 1.0f / (float)hw;
        inv_wg = 1.0f / (float)g.Wg;
    }
    // rows r0 + step*e, e = 0..N-1 -> (image, y, x) with one integer division
    template <int N, typename F>
    __device__ __forceinline__ void decode(int r0, int step, F&& f) const {
        const int n0 = r0 / hw;
        const int rem0 = r0 - n0 * hw;
#pragma unroll
        for (int e = 0; e < N; ++e) {
            int rem = rem0 + step * e;
            const int dn = small_div(rem, inv_hw);
            rem -= dn * hw;
            const int y = small_div(rem, inv_wg);
            f(e, n0 + dn, y, rem - y * wg);
        }
    }
    // pixel-major order: r = pix * nimg + image.  Exact integer division: one or eight per lane and tile, next to
    // the tile's hundreds of MFMAs.
    template <int N, typename F>
    __device__ __forceinline__ void decode_pm(int nimg, int r0, int step, F&& f) const {
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const int r = r0 + step * e;
            const int pix = (int)((unsigned)r / (unsigned)nimg);
            const int y = small_div(pix, inv_wg);               // pix < Hg*Wg + a few tiles' worth: far below 2^21
            f(e, r - pix * nimg, y, pix - y * wg);
        }
    }
};

// Taps tap0 .. tap0+ntaps-1 that are inside the image for at least one row of tile [row0, row0 + rows), as a bit mask
// (wave-uniform scalar code).  Image-major order, a tile that spans two grid positions, or a tile with no valid tap
// (image-major, or a tile that spans two grid positions): every tap (rows are masked individually either way; the mask
// only removes work that is zero for the whole tile).  A tile with no valid tap at all — a tap group of an accumulating
// data gradient at a border position — takes the first tap, fully masked: its epilogue still runs.
struct TileInfo { unsigned mask; };
template <bool PM>
__device__ __forceinline__ TileInfo tile_info(const ConvGeom& g, int row0, int rows, int tap0, int ntaps) {
    const unsigned full = ntaps >= 32 ? 0xffffffffu : ((1u << ntaps) - 1u);
    TileInfo ti{full};
    if (!PM) return ti;
    row0 = __builtin_amdgcn_readfirstlane(row0);
    if (row0 >= g.rows_total) { ti.mask = 1u; return ti; }
    int last = row0 + rows - 1;
    if (last >= g.rows_total) last = g.rows_total - 1;
    const int p0 = row0 / g.pm_nimg, p1 = last / g.pm_nimg;
    if (p0 != p1) return ti;
    const int y = p0 / g.Wg, x = p0 - y * g.Wg;
    const int ys = y * g.sy, xs = x * g.sx;
    unsigned m = 0;
    for (int t = 0; t < ntaps; ++t) {
        const bool in = ((unsigned)(ys + g.dy[tap0 + t]) < (unsigned)g.Hi) & ((unsigned)(xs + g.dx[tap0 + t]) < (unsigned)g.Wi);
        m |= in ? (1u << t) : 0u;
    }
    ti.mask = __builtin_amdgcn_readfirstlane(m ? m : 1u);
    return ti;
}
// The tap tables of a launch held in registers: lane t (mod 32) of a wave keeps tap t's offsets, a wave-uniform tap
// index fetches them with v_readlane (no scalar memory load and its lgkmcnt wait in front of every step's address
// arithmetic), and a tile's in-image tap mask is one compare per lane and a ballot instead of a scalar loop over taps.
struct TapTable {
    int dy, dx, col, woff;
    __device__ __forceinline__ TapTable(const ConvGeom& g, int lane) {
        const int t = lane & 31;
        dy = g.dy[t]; dx = g.dx[t]; col = g.a_col[t]; woff = g.w_off[t];
    }
    __device__ __forceinline__ int tdy(int t) const { return __builtin_amdgcn_readlane(dy, t); }
    __device__ __forceinline__ int tdx(int t) const { return __builtin_amdgcn_readlane(dx, t); }
    __device__ __forceinline__ int tcol(int t) const { return __builtin_amdgcn_readlane(col, t); }
    __device__ __forceinline__ int twoff(int t) const { return __builtin_amdgcn_readlane(woff, t); }
    // as tile_info<PM>() below, for taps tap0 .. tap0+ntaps-1 (called by all 64 lanes)
    template <bool PM>
    __device__ __forceinline__ TileInfo tile(const ConvGeom& g, int lane, int row0, int rows, int tap0, int ntaps) const {
        const unsigned full = ntaps >= 32 ? 0xffffffffu : ((1u << ntaps) - 1u);
        TileInfo ti{full};
        if (!PM) return ti;
        row0 = __builtin_amdgcn_readfirstlane(row0);
        if (row0 >= g.rows_total) { ti.mask = 1u; return ti; }
        int last = row0 + rows - 1;
        if (last >= g.rows_total) last = g.rows_total - 1;
        const int p0 = row0 / g.pm_nimg, p1 = last / g.pm_nimg;
        if (p0 != p1) return ti;
        const int y = p0 / g.Wg, x = p0 - y * g.Wg;
        const int tl = lane & 31;
        const bool in = (lane < 32) & (tl >= tap0) & (tl < tap0 + ntaps) &
                        ((unsigned)(y * g.sy + dy) < (unsigned)g.Hi) & ((unsigned)(x * g.sx + dx) < (unsigned)g.Wi);
        const unsigned m = (unsigned)__builtin_amdgcn_ballot_w64(in) >> tap0;
        ti.mask = m ? m : 1u;
        return ti;
    }
};

// lowest set bit at or above position `from` (mask has one): index; mask_above(m, t) = bits of m above t
__device__ __forceinline__ int first_tap(unsigned m) { return __builtin_ctz(m); }
__device__ __forceinline__ unsigned taps_above(unsigned m, int t) { return t >= 31 ? 0u : (m & ~((2u << t) - 1u)); }

__device__ __forceinline__ RowInfo make_row(const ConvGeom& g, int r, int n, int y, int x) {
    RowInfo ri;
    ri.ys = r < g.rows_total ? y * g.sy : INVALID_YS;
    ri.xs = x * g.sx;
    ri.off = ((n * g.Hi + y * g.sy) * g.Wi + ri.xs) * g.a_pitch;
    return ri;
}

__device__ __forceinline__ RowInfo row_info(const ConvGeom& g, int r) {
    const int hw = g.Hg * g.Wg;
    const int n = r / hw;
    const int rem = r - n * hw;
    const int y = rem / g.Wg;
    return make_row(g, r, n, y, rem - y * g.Wg);
}

__device__ __forceinline__ int out_pixel(const ConvGeom& g, int r) {
    if (r >= g.rows_total) return -1;
    const int hw = g.Hg * g.Wg;
    const int n = r / hw;
    const int rem = r - n * hw;
    const int y = rem / g.Wg;
    const int x = rem - y * g.Wg;
    return (n * g.Ho + y * g.oys + g.oy0) * g.Wo + x * g.oxs + g.ox0;
}
// Load 4 channels of the source pixel of row `ri` for tap t; *ok says whether the tap is inside the image.
// Branch-free: an outside tap loads element 0 (always mapped) and is zeroed by transform_a; a branch around
// each load would make hipcc wait for every load separately.  Tap quantities are wave-uniform (SGPRs).
__device__ __forceinline__ float4 load_a(const ConvGeom& g, const float* __restrict__ A, const RowInfo& ri, int t,
                                         int q, bool* ok) {
    const int dy = g.dy[t], dx = g.dx[t];
    const int delta = (dy * g.Wi + dx) * g.a_pitch + g.a_col[t];
    *ok = ((unsigned)(ri.ys + dy) < (unsigned)g.Hi) & ((unsigned)(ri.xs + dx) < (unsigned)g.Wi);
    const int off = *ok ? ri.off + delta : 0;
    return *reinterpret_cast<const float4*>(A + off + 4 * q);
}

// previous layer's BatchNorm + ReLU folded into the operand load; padding stays exactly zero.
// On this hardware every VALU instruction of a wave takes its 4-8 cycles out of the SIMD's MFMA time (nothing overlaps:
// tools/mfma_power.hip, 32 VALU ops next to 32 MFMAs cost 13 %), so the fold is written for the fewest instructions:
// v_pk_fma_f32 on channel pairs, one v_max per channel, and the zeroing of rows outside the image only on waves that
// have such a row (`fold_a` is the version for waves whose 64 rows are all inside).
typedef float floatx2 __attribute__((ext_vector_type(2)));
template <bool HAS_T>
__device__ __forceinline__ float4 fold_a(float4 v, const float4& sc, const float4& sh) {
    if (HAS_T) {
        floatx2 lo = {v.x, v.y}, hi = {v.z, v.w};
        lo = __builtin_elementwise_fma(lo, floatx2{sc.x, sc.y}, floatx2{sh.x, sh.y});
        hi = __builtin_elementwise_fma(hi, floatx2{sc.z, sc.w}, floatx2{sh.z, sh.w});
        v.x = fmaxf(lo.x, 0.f); v.y = fmaxf(lo.y, 0.f); v.z = fmaxf(hi.x, 0.f); v.w = fmaxf(hi.y, 0.f);
    }
    return v;
}
template <bool HAS_T>
__device__ __forceinline__ float4 transform_a(float4 v, bool ok, const float4& sc, const float4& sh) {
    v = fold_a<HAS_T>(v, sc, sh);
    v.x = ok ? v.x : 0.f;
    v.y = ok ? v.y : 0.f;
    v.z = ok ? v.z : 0.f;
    v.w = ok ? v.w : 0.f;
    return v;
}

// The same fold in ONE instruction per channel pair: v_pk_fma_f32 with the CLAMP modifier clamps its result to [0, 1],
// which is a ReLU as long as nothing reaches 1 — so the fold constants are scaled by 2^-60 (scale * 2^-60, shift * 2^-60:
// the folded activation comes out as relu(bn(y)) * 2^-60, exactly, powers of two commute with the rounding of the fma)
// and the other MFMA operand carries the 2^60 (weights staged into LDS as W * 2^60; the streaming weight gradient
// multiplies its accumulators by 2^60 once, at the end).  Products and sums are bit for bit those of the unscaled
// operands (tools/bin/clamp_test; an activation below 2^-66 would be flushed, far below fp32 resolution of any sum it
// enters).  16 VALU instructions per 32 operand values instead of 16 + 32 v_max: every VALU instruction next to the
// MFMAs costs ~5 cycles of the SIMD's MFMA time (tools/mfma_valu_cost.hip).
constexpr float FOLD_UP = 0x1p60f, FOLD_DOWN = 0x1p-60f;
__device__ __forceinline__ floatx2 pk_fma_clamp01(floatx2 a, floatx2 b, floatx2 c) {
    floatx2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// sc, sh: the fold constants times FOLD_DOWN; returns relu(v * scale + shift) * FOLD_DOWN (v itself without a fold)
template <bool HAS_T>
__device__ __forceinline__ float4 fold_a_scaled(float4 v, const float4& sc, const float4& sh) {
    if (HAS_T) {
        const floatx2 lo = pk_fma_clamp01(floatx2{v.x, v.y}, floatx2{sc.x, sc.y}, floatx2{sh.x, sh.y});
        const floatx2 hi = pk_fma_clamp01(floatx2{v.z, v.w}, floatx2{sc.z, sc.w}, floatx2{sh.z, sh.w});
        v = make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    return v;
}
// The same for a consumer that may follow at once: a v_mfma must not read a VGPR within two wait states of the VALU
// instruction that wrote it, and the compiler's hazard recognizer does not look into inline asm (it left one wait
// state; the first MFMA behind the fold then read the register's previous value: tools/bin/hazard_test gives 2 as the
// distance that works).  Both halves and the wait states are one asm block, correct wherever it is scheduled.
template <bool HAS_T>
__device__ __forceinline__ float4 fold_a_scaled_guarded(float4 v, const float4& sc, const float4& sh) {
    if (HAS_T) {
        floatx2 lo, hi;
        asm("v_pk_fma_f32 %0, %2, %3, %4 clamp\n\tv_pk_fma_f32 %1, %5, %6, %7 clamp\n\ts_nop 1"
            : "=&v"(lo), "=v"(hi)
            : "v"(floatx2{v.x, v.y}), "v"(floatx2{sc.x, sc.y}), "v"(floatx2{sh.x, sh.y}), "v"(floatx2{v.z, v.w}),
              "v"(floatx2{sc.z, sc.w}), "v"(floatx2{sh.z, sh.w}));
        v = make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    return v;
}
__device__ __forceinline__ float4 scaled4(const float4& v, float f) { return make_float4(v.x * f, v.y * f, v.z * f, v.w * f); }

// ---------------------------------------------------------------------------- fp32 products on the bf16 matrix pipe
// An fp32 value is the exact sum of three bf16 pieces, x = x1 + x2 + x3 (round-to-nearest pieces; every residual is
// exact in fp32).  a * b is then nine exact bf16 x bf16 products, and the six with i + j <= 4 carry everything above
// 2^-24 |a b| — fp32's own rounding.  v_mfma_f32_32x32x16_bf16 does 16 x the MACs per cycle of the fp32 MFMA, so the
// six cost 6 / 16 of it; the accumulation is fp32 as before.  Measured (tools/split_mfma_cost.hip, random operands,
// error against float64 in units of 2^-24 sum|a b|): mean 0.58 / max 7.5 this way, 0.74 / 8.2 with the fp32 MFMA — it
// is not a reduced-precision path.  The chip lowers its clock under the denser pipe, so the loop gains 1.5 x, not 2.7.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
struct Split3 { unsigned p1, p2, p3; };                  // two values per register: low half = the first one's piece
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
struct Frag3 { uint4 p[3]; };                            // 8 values: three planes of 8 bf16
__device__ __forceinline__ Frag3 split8(const float4& u, const float4& v, const SplitSel& q) {
    const Split3 a = split_pair(u.x, u.y, q), b = split_pair(u.z, u.w, q), c = split_pair(v.x, v.y, q), d = split_pair(v.z, v.w, q);
    Frag3 f;
    f.p[0] = make_uint4(a.p1, b.p1, c.p1, d.p1);
    f.p[1] = make_uint4(a.p2, b.p2, c.p2, d.p2);
    f.p[2] = make_uint4(a.p3, b.p3, c.p3, d.p3);
    return f;
}
__device__ __forceinline__ bf16x8 as_bf(const uint4& u) { return __builtin_bit_cast(bf16x8, u); }
__device__ __forceinline__ floatx16 mfma_split6(const Frag3& a, const Frag3& b, floatx16 c) {      // small terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[2]), as_bf(b.p[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[0]), as_bf(b.p[2]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[1]), as_bf(b.p[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[1]), as_bf(b.p[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[0]), as_bf(b.p[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a.p[0]), as_bf(b.p[0]), c, 0, 0, 0);
    return c;
}
// LDS image of one 64 x 64 weight tile for that path: three planes [64 co][64 ci] of bf16, 128-byte rows.  Channel
// c = 8 kk + 4 h + j (the lane's operand layout: float4 kk of lane half h) sits at element 16 (kk / 2) + 8 h + 4 (kk & 1)
// + j, so the 8 channels of MFMA step kk / 2 and lane half h are one 16-byte chunk; chunk q of row co is stored at chunk
// q ^ ((co >> 1) & 7), which makes the ds_read_b128 of 16 consecutive rows conflict-free without padding (six taps
// fit 160 KB: 147 456 bytes).
constexpr int BF3_TAP_BYTES = 3 * 64 * 128;
constexpr int BF3_TAPS = 6;                              // taps whose planes fit LDS at once
// rows from which a pixel-major weight-stationary launch runs its products on the bf16 pipe (BF3): from about three
// 256-row tiles per CU on, as for the 8-wave fp32 kernel
#ifndef CMVAE_DGRAD_PARITY
#define CMVAE_DGRAD_PARITY 1      // (0: a comparison build with the 16-tap data gradient's passes in tap order)
#endif
#ifndef CMVAE_BF3_LINEAR
#define CMVAE_BF3_LINEAR 16384     // rows from which nn.Linear launches take the bf16 pipe (0: never; see launch_gemm)
#endif
#ifndef CMVAE_BF3_MIN_ROWS
#define CMVAE_BF3_MIN_ROWS (3 * 256 * 256)     // (a comparison build sets it out of reach: tools/ab_bench.sh)
#endif
static int bf3_min_rows() { return CMVAE_BF3_MIN_ROWS; }
__device__ __forceinline__ int bf3_elem(int pl, int co, int chunk) {         // element (bf16) index within a tap
    return ((pl * 64 + co) * 8 + (chunk ^ ((co >> 1) & 7))) * 8;
}

template <int ACT>
__device__ __forceinline__ float apply_act(float v) {
    if (ACT == 1) return fmaxf(v, 0.f);
    if (ACT == 2) return elu1(v);
    return v;
}

// ---------------------------------------------------------------------------- forward gather-GEMM
// One software pipeline over the block's (tile, tap) steps: while the MFMAs of step s run out of LDS, the
// operands of step s+1 (possibly the first tap of the block's next tile) are already in flight to registers.
template <int ACT, bool HAS_T, bool LINEAR>
__global__ __launch_bounds__(256, 3) void conv_gemm_kernel(ConvGeom g, const float* __restrict__ A,
                                                           const float* __restrict__ Wt,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ a_scale,
                                                           const float* __restrict__ a_shift,
                                                           float* __restrict__ out, float* __restrict__ stats_part) {
    __shared__ __align__(16) float As[BM * LDA];
    __shared__ __align__(16) float Ws[64 * LDA];
    __shared__ int opix[2][BM];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int q = tid & 15, rb = tid >> 4;
    const int nb = blockIdx.y;
    const float* Wn = Wt + nb * g.w_nblk_stride + (long)rb * g.w_pitch + 4 * q;
    const long w16 = 16L * g.w_pitch;
    const int ntiles = (g.rows_total + BM - 1) / BM;
    const int nsteps = ((ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * g.ntaps;
    const RowDecoder dec(g);
    float ssum0 = 0.f, ssum1 = 0.f, ssq0 = 0.f, ssq1 = 0.f;
    const float b0 = bias ? bias[nb * 64 + i] : 0.f;
    const float b1 = bias ? bias[nb * 64 + 32 + i] : 0.f;

    RowInfo ri[8];
    float4 ra[8], rw0, rw1, rw2, rw3;
    bool ok[8];
    int ltile = blockIdx.x;          // tile whose operands are being loaded
    int par = 0;                     // opix buffer of the tile being multiplied
    auto rows_of_tile = [&](int tile, int* op) {
        if (LINEAR) {        // one row per "image": no (y, x) to decode (nn.Linear launches, own kernel symbol)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = tile * BM + rb + 16 * e;
                ri[e].off = r * g.a_pitch;
                ri[e].ys = r < g.rows_total ? 0 : INVALID_YS;
                ri[e].xs = 0;
            }
            if (tid < BM) op[tid] = tile * BM + tid < g.rows_total ? tile * BM + tid : -1;
        } else {
            dec.decode<8>(tile * BM + rb, 16, [&](int e, int n, int y, int x) {
                ri[e] = make_row(g, tile * BM + rb + 16 * e, n, y, x);
            });
            if (tid < BM) op[tid] = out_pixel(g, tile * BM + tid);
        }
    };
    rows_of_tile(ltile, opix[0]);
#pragma unroll
    for (int e = 0; e < 8; ++e) ra[e] = load_a(g, A, ri[e], 0, q, &ok[e]);
    {
        const float* Wn0 = Wn + g.w_off[0];
        rw0 = *reinterpret_cast<const float4*>(Wn0);
        rw1 = *reinterpret_cast<const float4*>(Wn0 + w16);
        rw2 = *reinterpret_cast<const float4*>(Wn0 + 2 * w16);
        rw3 = *reinterpret_cast<const float4*>(Wn0 + 3 * w16);
    }

    floatx16 acc0 = {}, acc1 = {};
    int t = 0;
    for (int s = 0; s < nsteps; ++s) {
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (HAS_T) {
            sc = *reinterpret_cast<const float4*>(a_scale + g.a_col[t] + 4 * q);
            sh = *reinterpret_cast<const float4*>(a_shift + g.a_col[t] + 4 * q);
        }
#if !(CMVAE_ABLATE & 2)
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e)
            *reinterpret_cast<float4*>(&As[(rb + 16 * e) * LDA + 4 * q]) = transform_a<HAS_T>(ra[e], ok[e], sc, sh);
        *reinterpret_cast<float4*>(&Ws[rb * LDA + 4 * q]) = rw0;
        *reinterpret_cast<float4*>(&Ws[(rb + 16) * LDA + 4 * q]) = rw1;
        *reinterpret_cast<float4*>(&Ws[(rb + 32) * LDA + 4 * q]) = rw2;
        *reinterpret_cast<float4*>(&Ws[(rb + 48) * LDA + 4 * q]) = rw3;
        __syncthreads();
#endif
        // operands of the next step (past the block's last step the rows are all invalid: element 0 is read)
        int tn = t + 1;
        if (tn == g.ntaps) {
            tn = 0;
            ltile += gridDim.x;
            rows_of_tile(ltile, opix[par ^ 1]);
        }
#if !(CMVAE_ABLATE & 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) ra[e] = load_a(g, A, ri[e], tn, q, &ok[e]);
        {
            const float* Wn1 = Wn + g.w_off[tn];
            rw0 = *reinterpret_cast<const float4*>(Wn1);
            rw1 = *reinterpret_cast<const float4*>(Wn1 + w16);
            rw2 = *reinterpret_cast<const float4*>(Wn1 + 2 * w16);
            rw3 = *reinterpret_cast<const float4*>(Wn1 + 3 * w16);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of the MFMAs that hide it
        const float* ap = &As[(32 * w + i) * LDA + 4 * h];
        const float* bp0 = &Ws[i * LDA + 4 * h];
        const float* bp1 = &Ws[(32 + i) * LDA + 4 * h];
        float4 a4 = *reinterpret_cast<const float4*>(ap);
        float4 p4 = *reinterpret_cast<const float4*>(bp0);
        float4 r4 = *reinterpret_cast<const float4*>(bp1);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float4 an = a4, pn = p4, rn = r4;
            if (kk < 7) {        // fragments of the next k-slice are read while this slice multiplies
                an = *reinterpret_cast<const float4*>(ap + 8 * kk + 8);
                pn = *reinterpret_cast<const float4*>(bp0 + 8 * kk + 8);
                rn = *reinterpret_cast<const float4*>(bp1 + 8 * kk + 8);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, p4.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, r4.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, p4.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, r4.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, p4.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, r4.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, p4.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, r4.w, acc1, 0, 0, 0);
            a4 = an; p4 = pn; r4 = rn;
        }
        if (tn == 0) {
            // epilogue of the finished tile: C[row = (reg&3) + 8*(reg>>2) + 4*h][col = i]
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const int op = opix[par][row];
                if (op >= 0) {
                    const float v0 = acc0[reg] + b0, v1 = acc1[reg] + b1;
                    ssum0 += v0; ssq0 = fmaf(v0, v0, ssq0);
                    ssum1 += v1; ssq1 = fmaf(v1, v1, ssq1);
                    float* o = out + (long)op * g.o_pitch + nb * 64;
                    o[i] = apply_act<ACT>(v0);
                    o[32 + i] = apply_act<ACT>(v1);
                }
            }
            acc0 = floatx16{};
            acc1 = floatx16{};
            par ^= 1;
        }
        t = tn;
    }
    if (stats_part) {
        ssum0 += __shfl_xor(ssum0, 32); ssum1 += __shfl_xor(ssum1, 32);
        ssq0 += __shfl_xor(ssq0, 32); ssq1 += __shfl_xor(ssq1, 32);
        float* red = As;     // [4 waves][128]
        __syncthreads();
        if (h == 0) {
            red[w * 128 + i] = ssum0; red[w * 128 + 32 + i] = ssum1;
            red[w * 128 + 64 + i] = ssq0; red[w * 128 + 96 + i] = ssq1;
        }
        __syncthreads();
        if (tid < 128) {
            const float v = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
            stats_part[((long)blockIdx.y * gridDim.x + blockIdx.x) * 128 + tid] = v;
        }
    }
}

// ---------------------------------------------------------------------------- forward, direct-operand variant
// For launches without the folded BatchNorm (data gradients, encoder convs, Linear): a lane owns ONE row of the
// tile and loads its operand directly in MFMA fragment layout (8 x 16 B: channels 8kk + 4h .. +3), one step ahead.
// The operand never passes through LDS: no staging stores, no A-fragment reads, one row decode per lane, and the
// two barriers of a step only guard the 16 KB weight tile.
// mse_target != NULL (nn.Linear launches of the feature models' last proj layer): the epilogue also accumulates the
// squared error of the activated output against target[row / mse_M][column] — the Gaussian likelihood of
// mimgnet/model.py:391 — into stats_part's column sums instead of the BatchNorm statistics.
// BNB (data-gradient launches of the decoder): the output is the gradient at a = relu(bn(x)); the epilogue reads x at
// the output's own pixels and leaves in stats_part the BatchNorm-backward sums of the layer below instead of the
// forward statistics: {sum g, sum g * xhat} per channel, g = out * (x*scale + shift > 0), xhat = (x - mean) * invstd
// (what bn_relu_bwd_reduce_kernel computes in a pass of its own over out and x).
struct BnBwd {
    const float* x;          // [pixels][o_pitch], the layout of out
    const float* scale; const float* shift; const float* mean; const float* invstd;     // [64] each
};
template <int ACT, bool LINEAR, bool PM, bool BNB = false>
__global__ __launch_bounds__(256, 3) void conv_gemm_direct_kernel(ConvGeom g, const float* __restrict__ A,
                                                                  const float* __restrict__ Wt,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ out,
                                                                  float* __restrict__ stats_part,
                                                                  const float* __restrict__ mse_target, int mse_M,
                                                                  int mse_pitch, BnBwd bnb = BnBwd{}) {
    __shared__ __align__(16) float Ws[64 * LDA];
    __shared__ int opix[2][BM];
    __shared__ float red[4 * 128];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int q = tid & 15, rb = tid >> 4;
    const int nb = blockIdx.y;
    const float* Wn = Wt + nb * g.w_nblk_stride + (long)rb * g.w_pitch + 4 * q;
    const long w16 = 16L * g.w_pitch;
    const int ntiles = (g.rows_total + BM - 1) / BM;
    const RowDecoder dec(g);
    float ssum0 = 0.f, ssum1 = 0.f, ssq0 = 0.f, ssq1 = 0.f;
    const float b0 = bias ? bias[nb * 64 + i] : 0.f;
    const float b1 = bias ? bias[nb * 64 + 32 + i] : 0.f;
    float bsc0 = 0.f, bsh0 = 0.f, bmu0 = 0.f, bis0 = 0.f, bsc1 = 0.f, bsh1 = 0.f, bmu1 = 0.f, bis1 = 0.f;
    if (BNB) {
        bsc0 = bnb.scale[i]; bsh0 = bnb.shift[i]; bmu0 = bnb.mean[i]; bis0 = bnb.invstd[i];
        bsc1 = bnb.scale[32 + i]; bsh1 = bnb.shift[32 + i]; bmu1 = bnb.mean[32 + i]; bis1 = bnb.invstd[32 + i];
    }

    RowInfo ri;
    // two operand register sets used alternately (the loop body is instantiated twice): the loads of step s+1 go
    // into the set the MFMAs of step s do not read, so no register copy sits between them
    float4 ra[8], rc[8], rw0, rw1, rw2, rw3;
    bool ok_a = false, ok_b = false;
    int par = 0;
    auto rows_of_tile = [&](int tile, const TileInfo& ti, int* op) {
        const int r = tile * BM + 32 * w + i;          // this lane's row
        if (LINEAR) {
            ri.off = r * g.a_pitch;
            ri.ys = r < g.rows_total ? 0 : INVALID_YS;
            ri.xs = 0;
            if (tid < BM) op[tid] = tile * BM + tid < g.rows_total ? tile * BM + tid : -1;
        } else if (PM) {
            dec.decode_pm<1>(g.pm_nimg, r, 0, [&](int, int n, int y, int x) { ri = make_row(g, r, n, y, x); });
            if (tid < BM) {
                const int ro = tile * BM + tid;
                int opx = -1;
                dec.decode_pm<1>(g.pm_nimg, ro, 0, [&](int, int n, int y, int x) {
                    if (ro < g.rows_total) opx = (n * g.Ho + y * g.oys + g.oy0) * g.Wo + x * g.oxs + g.ox0;
                });
                op[tid] = opx;
            }
        } else {
            dec.decode<1>(r, 0, [&](int, int n, int y, int x) { ri = make_row(g, r, n, y, x); });
            if (tid < BM) op[tid] = out_pixel(g, tile * BM + tid);
        }
    };
    const TapTable taps(g, lane);
    auto load_frag = [&](int t, float4 (&dst)[8], bool& ok_dst) {
        const int dy = taps.tdy(t), dx = taps.tdx(t);
        const int delta = (dy * g.Wi + dx) * g.a_pitch + taps.tcol(t);
        const bool ok = ((unsigned)(ri.ys + dy) < (unsigned)g.Hi) & ((unsigned)(ri.xs + dx) < (unsigned)g.Wi);
#if CMVAE_ABLATE == 11 || CMVAE_ABLATE == 13     // every operand load hits the same 8 KB
        const float* p = A + (ok ? (ri.off + delta) & 2047 : 0) + 4 * h;
#else
        const float* p = A + (ok ? ri.off + delta : 0) + 4 * h;
#endif
        ok_dst = ok;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) dst[kk] = *reinterpret_cast<const float4*>(p + 8 * kk);  // zeroed when consumed
    };
    auto load_w = [&](int t) {
        const float* Wn1 = Wn + taps.twoff(t);
        rw0 = *reinterpret_cast<const float4*>(Wn1);
        rw1 = *reinterpret_cast<const float4*>(Wn1 + w16);
        rw2 = *reinterpret_cast<const float4*>(Wn1 + 2 * w16);
        rw3 = *reinterpret_cast<const float4*>(Wn1 + 3 * w16);
    };
    // one software pipeline over the block's (tile, valid tap) steps
    int tile = blockIdx.x;
    TileInfo ti = taps.tile<PM>(g, lane, tile * BM, BM, 0, g.ntaps);
    unsigned mask = ti.mask;
    int t = first_tap(mask);
    rows_of_tile(tile, ti, opix[0]);
    load_frag(t, ra, ok_a);
    load_w(t);
    floatx16 acc0 = {}, acc1 = {};
    auto step = [&](float4 (&a)[8], bool& ok_cur, float4 (&nxt)[8], bool& ok_nxt) {
#if CMVAE_ABLATE != 14 && CMVAE_ABLATE != 15     // 14: no weight staging at all (timing only); 15: staging but no barriers
        __syncthreads();
#endif
#if CMVAE_ABLATE != 14
        *reinterpret_cast<float4*>(&Ws[rb * LDA + 4 * q]) = rw0;
        *reinterpret_cast<float4*>(&Ws[(rb + 16) * LDA + 4 * q]) = rw1;
        *reinterpret_cast<float4*>(&Ws[(rb + 32) * LDA + 4 * q]) = rw2;
        *reinterpret_cast<float4*>(&Ws[(rb + 48) * LDA + 4 * q]) = rw3;
#endif
        if (__builtin_amdgcn_ballot_w64(ok_cur) != ~0ull) {    // uniform: some row of the wave is outside the image
            const bool ok = ok_cur;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) a[kk] = transform_a<false>(a[kk], ok, a[kk], a[kk]);
        }
#if CMVAE_ABLATE != 14 && CMVAE_ABLATE != 15
        __syncthreads();
#endif
        const unsigned rest = taps_above(mask, t);
        const bool last = rest == 0;                   // uniform: the tile's last valid tap
        int tn, ntile = tile;
        unsigned nmask = mask;
        if (last) {
            ntile = tile + gridDim.x;
            ti = taps.tile<PM>(g, lane, ntile * BM, BM, 0, g.ntaps);
            nmask = ti.mask;
            tn = first_tap(nmask);
            rows_of_tile(ntile, ti, opix[par ^ 1]);    // past the block's last tile every row is invalid: element 0
        } else {
            tn = first_tap(rest);
        }
        load_frag(tn, nxt, ok_nxt);
#if CMVAE_ABLATE != 14
        load_w(tn);
#endif
        __builtin_amdgcn_sched_barrier(0);
        const float* bp0 = &Ws[i * LDA + 4 * h];
        const float* bp1 = &Ws[(32 + i) * LDA + 4 * h];
        float4 p4 = *reinterpret_cast<const float4*>(bp0);
        float4 r4 = *reinterpret_cast<const float4*>(bp1);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float4 pn = p4, rn = r4;
#if CMVAE_ABLATE != 14
            if (kk < 7) {
                pn = *reinterpret_cast<const float4*>(bp0 + 8 * kk + 8);
                rn = *reinterpret_cast<const float4*>(bp1 + 8 * kk + 8);
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            const float4 a4 = a[kk];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, p4.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, r4.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, p4.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, r4.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, p4.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, r4.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, p4.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, r4.w, acc1, 0, 0, 0);
            p4 = pn; r4 = rn;
        }
        if (last) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = 32 * w + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const int op = opix[par][row];
                if (op >= 0) {
                    const float v0 = acc0[reg] + b0, v1 = acc1[reg] + b1;
                    float* o = out + (long)op * g.o_pitch + nb * 64;
                    const float y0 = apply_act<ACT>(v0), y1 = apply_act<ACT>(v1);
                    o[i] = y0;
                    o[32 + i] = y1;
                    if (LINEAR && mse_target) {          // uniform
                        const float* tg = mse_target + (long)((unsigned)op / (unsigned)mse_M) * mse_pitch + nb * 64;
                        const float e0 = y0 - tg[i], e1 = y1 - tg[32 + i];
                        ssum0 = fmaf(e0, e0, ssum0);
                        ssum1 = fmaf(e1, e1, ssum1);
                    } else if (BNB) {
                        const float* xr = bnb.x + (long)op * g.o_pitch;
                        const float x0 = xr[i], x1 = xr[32 + i];
                        const float g0 = fmaf(x0, bsc0, bsh0) > 0.f ? v0 : 0.f;
                        const float g1 = fmaf(x1, bsc1, bsh1) > 0.f ? v1 : 0.f;
                        ssum0 += g0; ssq0 = fmaf(g0, (x0 - bmu0) * bis0, ssq0);
                        ssum1 += g1; ssq1 = fmaf(g1, (x1 - bmu1) * bis1, ssq1);
                    } else {
                        ssum0 += v0; ssq0 = fmaf(v0, v0, ssq0);
                        ssum1 += v1; ssq1 = fmaf(v1, v1, ssq1);
                    }
                }
            }
            acc0 = floatx16{};
            acc1 = floatx16{};
            par ^= 1;
            tile = ntile;
            mask = nmask;
        }
        t = tn;
    };
    while (tile < ntiles) {
        step(ra, ok_a, rc, ok_b);
        if (tile >= ntiles) break;
        step(rc, ok_b, ra, ok_a);
    }
    if (stats_part) {
        ssum0 += __shfl_xor(ssum0, 32); ssum1 += __shfl_xor(ssum1, 32);
        ssq0 += __shfl_xor(ssq0, 32); ssq1 += __shfl_xor(ssq1, 32);
        __syncthreads();
        if (h == 0) {
            red[w * 128 + i] = ssum0; red[w * 128 + 32 + i] = ssum1;
            red[w * 128 + 64 + i] = ssq0; red[w * 128 + 96 + i] = ssq1;
        }
        __syncthreads();
        if (tid < 128) {
            const float v = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
            stats_part[((long)blockIdx.y * gridDim.x + blockIdx.x) * 128 + tid] = v;
        }
    }
}

// ---------------------------------------------------------------------------- forward, weight-stationary variant
// The weight tiles of ALL taps of the launch sit in LDS for its whole duration and the operand comes straight from
// global memory in fragment layout (as in the direct kernel), with the folded BatchNorm + ReLU applied on the
// registers when there is one.  Nothing in the main loop is shared between waves, so it has NO block barrier: the
// waves of a workgroup drift apart and fill each other's load / epilogue bubbles.
//   NW = 4 waves, <= 4 taps (<= 68 KB of LDS, two workgroups per CU): the sub-pixel phases of the transposed convs;
//   NW = 8 waves, <= 9 taps (<= 157 KB, one workgroup per CU): 3x3 convolutions and their data gradients.
// Taps tap0 .. tap0+ntaps-1 of the geometry.
constexpr int WSTAT_TAPS4 = 4, WSTAT_TAPS8 = 9;
// bid / nblk: the workgroup's index among the workgroups that share this geometry (a launch may run several
// geometries side by side: conv_gemm_wstat_multi_kernel); stats_bid / stats_nblk: its index in the whole launch.
// ACCUM: the launch adds to what `out` already holds (the second half of a tap set too large for LDS).
template <int ACT, bool HAS_T, int NW, bool PM, bool BNB = false, bool ACCUM = false, bool BF3 = false>
__device__ __forceinline__ void wstat_body(
    const ConvGeom& g, const float* __restrict__ A, const float* __restrict__ Wt, const float* __restrict__ bias,
    const float* __restrict__ a_scale, const float* __restrict__ a_shift, float* __restrict__ out,
    float* __restrict__ stats_part, int stats_rows, int tap0, int ntaps, int bid, int nblk, int nb_in, int stats_bid,
    int stats_nblk, const BnBwd& bnb = BnBwd{}) {
    constexpr int BMT = 32 * NW;                                   // rows per tile: 32 per wave
#if CMVAE_ABLATE == 30
    if (threadIdx.x == 0 && stats_bid < 4096) g_wg_times[3 * stats_bid] = wall_clock64();
    int tiles_done = 0;
#endif
    extern __shared__ __align__(16) float wstat_smem[];
    float* Ws = wstat_smem;                                        // [ntaps][64][LDA]; BF3: [ntaps] bf16 plane images
    __bf16* Wp = reinterpret_cast<__bf16*>(wstat_smem);
    int* opix = BF3 ? reinterpret_cast<int*>(reinterpret_cast<char*>(wstat_smem) + (size_t)ntaps * BF3_TAP_BYTES)
                    : reinterpret_cast<int*>(Ws + ntaps * 64 * LDA);     // [2][NW waves][32 rows]
    float* red = reinterpret_cast<float*>(opix + 2 * NW * 32);     // [NW][128]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int q = tid & 15, rb = tid >> 4;
    const int nb = nb_in;
    const SplitSel ssel = make_split_sel();
    {
        const float* Wn = Wt + nb * g.w_nblk_stride + 4 * q;
        for (int t = 0; t < ntaps; ++t)
            for (int row = rb; row < 64; row += 4 * NW) {
                const float4 wv = scaled4(*reinterpret_cast<const float4*>(Wn + g.w_off[tap0 + t] + (long)row * g.w_pitch),
                                          HAS_T ? FOLD_UP : 1.f);    // the folded operand arrives times 2^-60 (fold_a_scaled)
                if (BF3) {
                    // channels 4q .. 4q+3 = float4 kk = q / 2 of lane half q & 1: chunk 2 (kk / 2) + (q & 1), elements
                    // 4 (kk & 1) .. + 3 of it
                    const Split3 s0 = split_pair(wv.x, wv.y, ssel), s1 = split_pair(wv.z, wv.w, ssel);
                    const int kk = q >> 1, chunk = 2 * (kk >> 1) + (q & 1), sub = 4 * (kk & 1);
                    __bf16* wt = Wp + (size_t)t * (BF3_TAP_BYTES / 2) + sub;
                    *reinterpret_cast<uint2*>(wt + bf3_elem(0, row, chunk)) = make_uint2(s0.p1, s1.p1);
                    *reinterpret_cast<uint2*>(wt + bf3_elem(1, row, chunk)) = make_uint2(s0.p2, s1.p2);
                    *reinterpret_cast<uint2*>(wt + bf3_elem(2, row, chunk)) = make_uint2(s0.p3, s1.p3);
                } else {
                    *reinterpret_cast<float4*>(&Ws[(t * 64 + row) * LDA + 4 * q]) = wv;
                }
            }
    }
    // BatchNorm fold of this lane's channels (8kk + 4h .. +3)
    float4 sc[HAS_T ? 8 : 1], sh[HAS_T ? 8 : 1];
    if (HAS_T) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            sc[kk] = scaled4(*reinterpret_cast<const float4*>(a_scale + 8 * kk + 4 * h), FOLD_DOWN);
            sh[kk] = scaled4(*reinterpret_cast<const float4*>(a_shift + 8 * kk + 4 * h), FOLD_DOWN);
        }
    }
    const int ntiles = (g.rows_total + BMT - 1) / BMT;
    const RowDecoder dec(g);
    float ssum0 = 0.f, ssum1 = 0.f, ssq0 = 0.f, ssq1 = 0.f;
    floatx2 sum0p = {0.f, 0.f}, sum1p = sum0p, sq0p = sum0p, sq1p = sum0p;      // fast epilogue: sums over even / odd rows
    const float b0 = bias ? bias[nb * 64 + i] : 0.f;
    const float b1 = bias ? bias[nb * 64 + 32 + i] : 0.f;
    float bsc0 = 0.f, bsh0 = 0.f, bmu0 = 0.f, bis0 = 0.f, bsc1 = 0.f, bsh1 = 0.f, bmu1 = 0.f, bis1 = 0.f;
    if (BNB) {
        bsc0 = bnb.scale[i]; bsh0 = bnb.shift[i]; bmu0 = bnb.mean[i]; bis0 = bnb.invstd[i];
        bsc1 = bnb.scale[32 + i]; bsh1 = bnb.shift[32 + i]; bmu1 = bnb.mean[32 + i]; bis1 = bnb.invstd[32 + i];
    }
    // uniform: every output byte offset fits 32 bits and every pixel index 24 (v_mad_u32_u24)
    const bool out_off32 = (long)(g.rows_total / (g.Hg * g.Wg)) * g.Ho * g.Wo < (1L << 24) &&
                           (long)(g.rows_total / (g.Hg * g.Wg)) * g.Ho * g.Wo * g.o_pitch * 4 < (1L << 32) &&
                           (long)g.o_pitch * 4 < (1L << 24);
    // (non-temporal stores of outputs far beyond L2, and loads of an adding pass's old output, as in common.h's
    // load_once / store_once: measured, no change of the step — 11.04 / 11.00 / 10.90 / 10.82 against 11.03 / 10.99 /
    // 10.85 / 10.84 ms)
    __syncthreads();                                               // the only barrier before the statistics

    RowInfo ri;
    float4 ra[8];
    bool ok_next = false;
    int par = 0;
    auto rows_of_tile = [&](int tile, const TileInfo& ti, int* op) {      // op: this wave's 32 output pixels
        const int r = tile * BMT + 32 * w + i;                     // this lane's row
        int opx = -1;
        auto set = [&](int, int n, int y, int x) {
            ri = make_row(g, r, n, y, x);
            if (r < g.rows_total) opx = (n * g.Ho + y * g.oys + g.oy0) * g.Wo + x * g.oxs + g.ox0;
        };
        if (PM) dec.decode_pm<1>(g.pm_nimg, r, 0, set);
        else dec.decode<1>(r, 0, set);
        if (h == 0) op[i] = opx;
    };
    const TapTable taps(g, lane);
    auto load_frag = [&](int t) {                                  // raw loads: fold / zeroing applied when consumed
        const int dy = taps.tdy(tap0 + t), dx = taps.tdx(tap0 + t);
        const int delta = (dy * g.Wi + dx) * g.a_pitch + taps.tcol(tap0 + t);
        ok_next = ((unsigned)(ri.ys + dy) < (unsigned)g.Hi) & ((unsigned)(ri.xs + dx) < (unsigned)g.Wi);
#if CMVAE_ABLATE == 11 || CMVAE_ABLATE == 13     // every operand load hits the same 8 KB
        const float* p = A + (ok_next ? (ri.off + delta) & 2047 : 0) + 4 * h;
#else
        const float* p = A + (ok_next ? ri.off + delta : 0) + 4 * h;
#endif
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ra[kk] = *reinterpret_cast<const float4*>(p + 8 * kk);
    };
    // Tile order.  Plain: tile = bid, bid + nblk, ...  Pixel-major launches (a tile = BMT images at ONE position, so
    // the taps of neighbouring positions and of the other sub-pixel phases gather the same operand rows) deal the
    // tiles in image-group-major order v = group * npix + position, one contiguous eighth of that order per XCD
    // (workgroup ids go round the 8 XCDs; each XCD has its own 4 MB L2): the workgroups of an XCD then work, at any
    // moment, on neighbouring positions of one image group — a working set of a few operand positions that its L2
    // holds — instead of on one position of many groups, where every gathered row was an L2 miss.
    const int gimg = PM ? g.pm_nimg / BMT : 0;                    // image groups
    const int gimg_d = gimg > 0 ? gimg : 1;                       // (divisor where gimg > 0 has been tested)
    // (not for images of few positions: when the XCD's stride nblk / 8 is a multiple of the positions per image, a
    // workgroup meets the same position in every group, and border positions have fewer taps than inner ones — the
    // 16-position data gradient of layer 2 took 400 us instead of 310; rotating the positions per group cured that
    // and cost the 49-position layers 10 %)
    const bool swz = PM && g.xcd_order && gridDim.y == 1 && (nblk & 7) == 0 && ((stats_bid - bid) & 7) == 0 &&
                     gimg > 0 && g.pm_nimg % BMT == 0 && ntiles % gimg_d == 0 && ntiles / gimg_d >= 32 &&
                     (nblk >> 3) % (ntiles / gimg_d) != 0;
    const int npos = swz ? ntiles / gimg_d : 1;
    int v = bid, v_end = ntiles, v_step = nblk;
    if (swz) {
        const int x = bid & 7;
        v = (int)((long)ntiles * x / 8) + (bid >> 3);
        v_end = (int)((long)ntiles * (x + 1) / 8);
        v_step = nblk >> 3;
    }
    auto tile_of = [&](int vv) {
        if (vv >= v_end) return ntiles;                            // past the end: the loop stops, the decoders cope
        if (!swz) return vv;
        const int grp = vv / npos;
        return (vv - grp * npos) * gimg + grp;
    };
#if CMVAE_ABLATE == 30
    const int my_tiles = v < v_end ? (v_end - v + v_step - 1) / v_step : 0;      // this workgroup's share of the tiles
#endif
    int tile = tile_of(v);
    TileInfo ti = taps.tile<PM>(g, lane, tile * BMT, BMT, tap0, ntaps);
    unsigned mask = ti.mask;
    int t = first_tap(mask);
    rows_of_tile(tile, ti, opix + w * 32);
    load_frag(t);
    floatx16 acc0 = {}, acc1 = {};
    // ACCUM: what the output holds is requested at the tile's first tap, not in the epilogue (where the round trip of
    // 32 loads per lane stood between the last MFMA and the stores of every tile: the layer-3 data gradient's adding
    // passes 436 -> 365-389 us).  (The BatchNorm-backward epilogue's x requested under the last tap's MFMAs as well:
    // 256 registers, 462 -> 452 us, nothing in the step time; it is loaded in the epilogue.)
    float old0[ACCUM ? 16 : 1], old1[ACCUM ? 16 : 1];
    bool old_here = false;                                         // uniform
    while (tile < ntiles) {
        float4 a[8];
        if (ACCUM && t == first_tap(mask)) {
            old_here = tile * BMT + 32 * w + 31 < g.rows_total && out_off32;
            if (old_here) {
                const int* op_w = opix + (par * NW + w) * 32;
                const unsigned col_b = (unsigned)(nb * 64 + i) * 4u, pitch_b = (unsigned)g.o_pitch * 4u;
                const char* ob = reinterpret_cast<const char*>(out);
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const int4 ops = *reinterpret_cast<const int4*>(&op_w[8 * rq + 4 * h]);
                    const int opv[4] = {ops.x, ops.y, ops.z, ops.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float* p0 = (const float*)(ob + (__umul24((unsigned)opv[r], pitch_b) + col_b));
                        old0[ACCUM ? 4 * rq + r : 0] = p0[0];
                        old1[ACCUM ? 4 * rq + r : 0] = p0[32];
                    }
                }
            }
        }
        {
            const bool ok = ok_next;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) a[kk] = fold_a_scaled<HAS_T>(ra[kk], sc[HAS_T ? kk : 0], sh[HAS_T ? kk : 0]);
            if (__builtin_amdgcn_ballot_w64(ok) != ~0ull) {        // uniform; rare: some row of the wave is outside the image
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    a[kk].x = ok ? a[kk].x : 0.f; a[kk].y = ok ? a[kk].y : 0.f;
                    a[kk].z = ok ? a[kk].z : 0.f; a[kk].w = ok ? a[kk].w : 0.f;
                }
            }
        }
        const unsigned rest = taps_above(mask, t);
        const bool last = rest == 0;                               // uniform: the tile's last valid tap
        int tn, ntile = tile;
        unsigned nmask = mask;
        if (last) {
            v += v_step;
            ntile = tile_of(v);
            ti = taps.tile<PM>(g, lane, ntile * BMT, BMT, tap0, ntaps);
            nmask = ti.mask;
            tn = first_tap(nmask);
            rows_of_tile(ntile, ti, opix + ((par ^ 1) * NW + w) * 32);
        } else {
            tn = first_tap(rest);
        }
        if (BF3) {
            Frag3 fa[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fa[ks] = split8(a[2 * ks], a[2 * ks + 1], ssel);
            load_frag(tn);
            __builtin_amdgcn_sched_barrier(0);
            const __bf16* wt = Wp + (size_t)t * (BF3_TAP_BYTES / 2);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                Frag3 b0, b1;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    b0.p[pl] = *reinterpret_cast<const uint4*>(wt + bf3_elem(pl, i, 2 * ks + h));
                    b1.p[pl] = *reinterpret_cast<const uint4*>(wt + bf3_elem(pl, 32 + i, 2 * ks + h));
                }
                acc0 = mfma_split6(fa[ks], b0, acc0);
                acc1 = mfma_split6(fa[ks], b1, acc1);
            }
        } else {
        load_frag(tn);
        __builtin_amdgcn_sched_barrier(0);
        const float* bp0 = &Ws[(t * 64 + i) * LDA + 4 * h];
        const float* bp1 = bp0 + 32 * LDA;
        float4 p4 = *reinterpret_cast<const float4*>(bp0);
        float4 r4 = *reinterpret_cast<const float4*>(bp1);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float4 pn = p4, rn = r4;
            if (kk < 7) {
                pn = *reinterpret_cast<const float4*>(bp0 + 8 * kk + 8);
                rn = *reinterpret_cast<const float4*>(bp1 + 8 * kk + 8);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float4 a4 = a[kk];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, p4.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, r4.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, p4.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, r4.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, p4.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, r4.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, p4.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, r4.w, acc1, 0, 0, 0);
            p4 = pn; r4 = rn;
        }
        }
        if (last && tile * BMT + 32 * w + 31 < g.rows_total && out_off32) {
            // every row of the wave exists and the output fits 32-bit byte offsets (the common case): no per-row
            // predicate, packed bias / statistics arithmetic on row pairs, scalar base + one VGPR offset per store
            const int* op_w = opix + (par * NW + w) * 32;
            const floatx2 bb0 = {b0, b0}, bb1 = {b1, b1};
            const unsigned col_b = (unsigned)(nb * 64 + i) * 4u, pitch_b = (unsigned)g.o_pitch * 4u;
            const char* ob = reinterpret_cast<const char*>(out);
            const char* xb = reinterpret_cast<const char*>(bnb.x);
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int4 ops = *reinterpret_cast<const int4*>(&op_w[8 * rq + 4 * h]);
                const int opv[4] = {ops.x, ops.y, ops.z, ops.w};
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    floatx2 v0 = floatx2{acc0[4 * rq + r], acc0[4 * rq + r + 1]} + bb0;
                    floatx2 v1 = floatx2{acc1[4 * rq + r], acc1[4 * rq + r + 1]} + bb1;
                    if (ACCUM) {
                        v0 += floatx2{old0[ACCUM ? 4 * rq + r : 0], old0[ACCUM ? 4 * rq + r + 1 : 0]};
                        v1 += floatx2{old1[ACCUM ? 4 * rq + r : 0], old1[ACCUM ? 4 * rq + r + 1 : 0]};
                    }
                    if (BNB) {
                        const float* x0 = (const float*)(xb + (__umul24((unsigned)opv[r], pitch_b) + col_b));
                        const float* x1 = (const float*)(xb + (__umul24((unsigned)opv[r + 1], pitch_b) + col_b));
                        const floatx2 xa = {x0[0], x1[0]}, xc = {x0[32], x1[32]};
                        const floatx2 ma = __builtin_elementwise_fma(xa, floatx2{bsc0, bsc0}, floatx2{bsh0, bsh0});
                        const floatx2 mc = __builtin_elementwise_fma(xc, floatx2{bsc1, bsc1}, floatx2{bsh1, bsh1});
                        const floatx2 g0 = {ma.x > 0.f ? v0.x : 0.f, ma.y > 0.f ? v0.y : 0.f};
                        const floatx2 g1 = {mc.x > 0.f ? v1.x : 0.f, mc.y > 0.f ? v1.y : 0.f};
                        sum0p += g0; sum1p += g1;
                        sq0p = __builtin_elementwise_fma(g0, (xa - floatx2{bmu0, bmu0}) * floatx2{bis0, bis0}, sq0p);
                        sq1p = __builtin_elementwise_fma(g1, (xc - floatx2{bmu1, bmu1}) * floatx2{bis1, bis1}, sq1p);
                    } else {
                        sum0p += v0; sq0p = __builtin_elementwise_fma(v0, v0, sq0p);
                        sum1p += v1; sq1p = __builtin_elementwise_fma(v1, v1, sq1p);
                    }
#if !(CMVAE_ABLATE == 12 || CMVAE_ABLATE == 13)
                    float* o0 = (float*)(ob + (__umul24((unsigned)opv[r], pitch_b) + col_b));
                    float* o1 = (float*)(ob + (__umul24((unsigned)opv[r + 1], pitch_b) + col_b));
                    o0[0] = apply_act<ACT>(v0.x); o0[32] = apply_act<ACT>(v1.x);
                    o1[0] = apply_act<ACT>(v0.y); o1[32] = apply_act<ACT>(v1.y);
#endif
                }
            }
            acc0 = floatx16{};
            acc1 = floatx16{};
            par ^= 1;
#if CMVAE_ABLATE == 30
            ++tiles_done;
            if (threadIdx.x == 0 && stats_bid < 4096) {
                if (tiles_done == my_tiles / 4) g_wg_quarters[4 * stats_bid] = wall_clock64();
                if (tiles_done == my_tiles / 2) g_wg_quarters[4 * stats_bid + 1] = wall_clock64();
                if (tiles_done == 3 * my_tiles / 4) g_wg_quarters[4 * stats_bid + 2] = wall_clock64();
                if (tiles_done == 7 * my_tiles / 8) g_wg_quarters[4 * stats_bid + 3] = wall_clock64();
            }
#endif
            tile = ntile;
            mask = nmask;
        } else if (last) {
            const int* op_w = opix + (par * NW + w) * 32;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int op = op_w[(reg & 3) + 8 * (reg >> 2) + 4 * h];
                if (op >= 0) {
                    float* o = out + (long)op * g.o_pitch + nb * 64;
                    float v0 = acc0[reg] + b0, v1 = acc1[reg] + b1;
                    if (ACCUM) { v0 += o[i]; v1 += o[32 + i]; }
                    if (BNB) {
                        const float* xr = bnb.x + (long)op * g.o_pitch + nb * 64;
                        const float x0 = xr[i], x1 = xr[32 + i];
                        const float g0 = fmaf(x0, bsc0, bsh0) > 0.f ? v0 : 0.f;
                        const float g1 = fmaf(x1, bsc1, bsh1) > 0.f ? v1 : 0.f;
                        ssum0 += g0; ssq0 = fmaf(g0, (x0 - bmu0) * bis0, ssq0);
                        ssum1 += g1; ssq1 = fmaf(g1, (x1 - bmu1) * bis1, ssq1);
                    } else {
                        ssum0 += v0; ssq0 = fmaf(v0, v0, ssq0);
                        ssum1 += v1; ssq1 = fmaf(v1, v1, ssq1);
                    }
#if CMVAE_ABLATE == 12 || CMVAE_ABLATE == 13     // no output stores (the statistics keep the accumulators alive)
                    if (v0 == 1.2345e33f) o[i] = v1;
#else
                    o[i] = apply_act<ACT>(v0);
                    o[32 + i] = apply_act<ACT>(v1);
#endif
                }
            }
            acc0 = floatx16{};
            acc1 = floatx16{};
            par ^= 1;
#if CMVAE_ABLATE == 30
            ++tiles_done;
            if (threadIdx.x == 0 && stats_bid < 4096) {
                if (tiles_done == my_tiles / 4) g_wg_quarters[4 * stats_bid] = wall_clock64();
                if (tiles_done == my_tiles / 2) g_wg_quarters[4 * stats_bid + 1] = wall_clock64();
                if (tiles_done == 3 * my_tiles / 4) g_wg_quarters[4 * stats_bid + 2] = wall_clock64();
                if (tiles_done == 7 * my_tiles / 8) g_wg_quarters[4 * stats_bid + 3] = wall_clock64();
            }
#endif
            tile = ntile;
            mask = nmask;
        }
        t = tn;
    }
#if CMVAE_ABLATE == 30
    if (threadIdx.x == 0 && stats_bid < 4096) {
        g_wg_times[3 * stats_bid + 1] = wall_clock64();
        g_wg_times[3 * stats_bid + 2] = tiles_done;
    }
#endif
    if (stats_part) {
        ssum0 += sum0p.x + sum0p.y; ssum1 += sum1p.x + sum1p.y;
        ssq0 += sq0p.x + sq0p.y; ssq1 += sq1p.x + sq1p.y;
        ssum0 += __shfl_xor(ssum0, 32); ssum1 += __shfl_xor(ssum1, 32);
        ssq0 += __shfl_xor(ssq0, 32); ssq1 += __shfl_xor(ssq1, 32);
        if (h == 0) {
            red[w * 128 + i] = ssum0; red[w * 128 + 32 + i] = ssum1;
            red[w * 128 + 64 + i] = ssq0; red[w * 128 + 96 + i] = ssq1;
        }
        __syncthreads();
        if (tid < 128) {
            float v = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) v += red[ww * 128 + tid];
            stats_part[(long)stats_bid * 128 + tid] = v;
            // the caller sized the partials for cmvae_conv_gemm_grid() workgroups: the rows this launch does not
            // own (it runs fewer, one or two per CU) are zeroed
            for (int r = stats_bid + stats_nblk; r < stats_rows; r += stats_nblk) stats_part[(long)r * 128 + tid] = 0.f;
        }
    }
}

template <int ACT, bool HAS_T, int NW, bool PM, bool BNB = false, bool ACCUM = false, bool BF3 = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv_gemm_wstat_kernel(
    ConvGeom g, const float* __restrict__ A, const float* __restrict__ Wt, const float* __restrict__ bias,
    const float* __restrict__ a_scale, const float* __restrict__ a_shift, float* __restrict__ out,
    float* __restrict__ stats_part, int stats_rows, int tap0, int ntaps, BnBwd bnb = BnBwd{}) {
    wstat_body<ACT, HAS_T, NW, PM, BNB, ACCUM, BF3>(g, A, Wt, bias, a_scale, a_shift, out, stats_part, stats_rows, tap0, ntaps,
                                        (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y, (int)blockIdx.x,
                                        (int)gridDim.x, bnb);
}

// Up to four geometries over the same operand, weight and output tensors in ONE launch — the sub-pixel phases of a
// transposed convolution.  Every workgroup belongs to one geometry (begin[] partitions the grid in proportion to the
// phases' tap work) and keeps that phase's <= 4 weight tiles in LDS.  One launch instead of four saves the fill, ramp
// and drain of three launches, which is most of the time of the small layers' phases.
struct ConvGeomMulti {
    ConvGeom g[4];
    int begin[5];
};
// BF3: the products run on the bf16 pipe (mfma_split6); 8 waves, one workgroup per CU (its four taps take 98 KB).
template <int ACT, bool HAS_T, bool PM, bool BF3 = false>
__global__ __launch_bounds__(BF3 ? 512 : 256, BF3 ? 1 : 2) void conv_gemm_wstat_multi_kernel(
    ConvGeomMulti p, const float* __restrict__ A, const float* __restrict__ Wt, const float* __restrict__ bias,
    const float* __restrict__ a_scale, const float* __restrict__ a_shift, float* __restrict__ out,
    float* __restrict__ stats_part, int stats_rows) {
    const int b = blockIdx.x;
    const int ph = (b >= p.begin[1]) + (b >= p.begin[2]) + (b >= p.begin[3]);          // uniform
    const ConvGeom& g = p.g[ph];
    wstat_body<ACT, HAS_T, BF3 ? 8 : 4, PM, false, false, BF3>(g, A, Wt, bias, a_scale, a_shift, out, stats_part,
                                                               stats_rows, 0, g.ntaps, b - p.begin[ph],
                                                               p.begin[ph + 1] - p.begin[ph], 0, b, (int)gridDim.x);
}

template <int ACT, bool HAS_T, int NW, bool PM, bool BNB = false, bool ACCUM = false, bool BF3 = false>
static int launch_wstat(const ConvGeom& g, const float* A, const float* Wt, const float* bias, const float* a_scale,
                        const float* a_shift, float* out, float* stats_part, int stats_rows, int n_col_blocks,
                        int tap0, int ntaps, hipStream_t st, BnBwd bnb = BnBwd{}) {
    const size_t lds = (BF3 ? (size_t)ntaps * BF3_TAP_BYTES : (size_t)ntaps * 64 * LDA * sizeof(float)) +
                       2 * NW * 32 * sizeof(int) + NW * 128 * sizeof(float);
    static const bool opted = [] {
        (void)hipFuncSetAttribute((const void*)conv_gemm_wstat_kernel<ACT, HAS_T, NW, PM, BNB, ACCUM, BF3>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        return true;
    }();
    (void)opted;
    const int ntiles = (g.rows_total + 32 * NW - 1) / (32 * NW);
    // resident workgroups: two (NW = 4) or one (NW = 8) per CU
    static const int cap = resident_blocks((const void*)conv_gemm_wstat_kernel<ACT, HAS_T, NW, PM, BNB, ACCUM, BF3>, 64 * NW,
                                           (NW == 4 ? 2 : 1) * compute_units());
    // (nn.Linear launches: the column blocks share the resident workgroups)
    int most = (NW == 4 ? 2 : 1) * compute_units() / (n_col_blocks > 0 ? n_col_blocks : 1);
    if (most < 1) most = 1;
    const int cap_use = cap < most ? cap : most;
    const dim3 grid(ntiles < cap_use ? ntiles : cap_use, n_col_blocks);
    hipLaunchKernelGGL((conv_gemm_wstat_kernel<ACT, HAS_T, NW, PM, BNB, ACCUM, BF3>), grid, dim3(64 * NW), lds, st, g, A, Wt, bias,
                       a_scale, a_shift, out, stats_part, stats_rows, tap0, ntaps, bnb);
    return check_launch("conv_gemm_wstat_kernel");
}

// Tap groups of a launch with more taps than fit LDS as bf16 planes (BF3_TAPS): gp = g with its taps ordered by
// stride-2 parity class where the geometry has them (a pass then reads its part of a gathered tensor once), cut[p] ..
// cut[p + 1] - 1 = the taps of pass p; returns the number of passes (<= 4).
static int bf3_pass_plan(const ConvGeom& g, ConvGeom& gp, int cut[8]) {
    gp = g;
    int ncut = 0;
    cut[0] = 0;
    bool by_class = false;
#if CMVAE_DGRAD_PARITY
    if (g.sy == 2 && g.sx == 2) {
        int order[32], cnt[4] = {0, 0, 0, 0}, n = 0;
        for (int c = 3; c >= 0; --c)                              // (odd, odd) first: the largest class of a 3 x 3 kernel
            for (int t = 0; t < g.ntaps; ++t)
                if (((g.dy[t] & 1) * 2 + (g.dx[t] & 1)) == c) { order[n++] = t; ++cnt[c]; }
        by_class = cnt[0] <= BF3_TAPS && cnt[1] <= BF3_TAPS && cnt[2] <= BF3_TAPS && cnt[3] <= BF3_TAPS;
        if (by_class) {
            for (int t = 0; t < g.ntaps; ++t) {
                gp.dy[t] = g.dy[order[t]]; gp.dx[t] = g.dx[order[t]];
                gp.a_col[t] = g.a_col[order[t]]; gp.w_off[t] = g.w_off[order[t]];
            }
            int in_pass = 0, pos = 0;
            for (int c = 3; c >= 0; --c) {
                if (cnt[c] == 0) continue;
                if (in_pass + cnt[c] > BF3_TAPS) { cut[++ncut] = pos; in_pass = 0; }
                in_pass += cnt[c]; pos += cnt[c];
            }
            cut[++ncut] = g.ntaps;
        }
    }
#endif
    if (!by_class) {
        ncut = (g.ntaps + BF3_TAPS - 1) / BF3_TAPS;
        for (int p = 1; p <= ncut; ++p) cut[p] = (int)((long)g.ntaps * p / ncut);
    }
    return ncut;
}

template <int ACT, bool HAS_T, bool PM>
static int dispatch_wstat(const ConvGeom& g, const float* A, const float* Wt, const float* bias, const float* a_scale,
                          const float* a_shift, float* out, float* stats_part, int stats_rows, int n_col_blocks,
                          hipStream_t st, bool* handled) {
    *handled = true;
    if (g.ntaps <= WSTAT_TAPS4)
        return launch_wstat<ACT, HAS_T, 4, PM>(g, A, Wt, bias, a_scale, a_shift, out, stats_part, stats_rows,
                                               n_col_blocks, 0, g.ntaps, st);
    // 8 waves, one workgroup per CU: worth its 157 KB weight fill from a few tiles per workgroup on.  (Splitting a
    // 16-tap data gradient into two accumulating 8-tap launches was measured: no gain, the read-modify-write of the
    // output eats what the barriers cost.)
    // (below three 256-row tiles per workgroup its single round of 256 workgroups is mostly tail: the few-shot
    // evaluation's second encoder layer, 78400 rows = 306 tiles, ran 125 us here against the direct kernel's 128-row
    // tiles on three workgroups per CU)
    // (a launch of at most one 256-row tile per CU — a single round again — was measured too: no gain, 13.07 vs 13.07 ms)
    if (g.ntaps <= WSTAT_TAPS8 && g.rows_total >= 3 * 256 * 256) {
        // 7 .. 9 taps (the encoder's 3 x 3 convs and their data gradients) on the bf16 pipe (mfma_split6): two passes of at
        // most six taps of weight planes, the second adding to the first one's output and carrying the bias and the
        // statistics.  Same-box A/B of the Omniglot 20-way step: 11.41 / 11.33 -> 11.27 / 11.21 ms.  (Stride-2 data
        // gradients with the BatchNorm-backward sums go through cmvae_conv_gemm_bnbwd, which cuts its passes at the taps'
        // parity classes.)
        if (ACT == 0 && n_col_blocks == 1 && g.ntaps > BF3_TAPS && g.rows_total >= bf3_min_rows()) {
            const int t1 = (g.ntaps + 1) / 2;
            if (int e = launch_wstat<0, HAS_T, 8, PM, false, false, true>(g, A, Wt, nullptr, a_scale, a_shift, out, nullptr, 0,
                                                                          1, 0, t1, st)) return e;
            return launch_wstat<0, HAS_T, 8, PM, false, true, true>(g, A, Wt, bias, a_scale, a_shift, out, stats_part,
                                                                    stats_rows, 1, t1, g.ntaps - t1, st);
        }
        return launch_wstat<ACT, HAS_T, 8, PM>(g, A, Wt, bias, a_scale, a_shift, out, stats_part, stats_rows,
                                               n_col_blocks, 0, g.ntaps, st);
    }
    *handled = false;
    return 0;
}

// ---------------------------------------------------------------------------- weight gradient
// part[blk][t][ci][co] = sum_r T(A[src(r,t)])[ci] * G[outpix(r)][g_col + co]
// Same software pipeline as the forward kernel: while the 64 MFMAs of a tap run, the next tap's operand rows (or
// the next tile's gradient rows and first tap) are in flight; LDS fragments are read four row-pairs ahead.
template <int NT, bool HAS_T, bool PM>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(ConvGeom g, int tap0, const float* __restrict__ A,
                                                            const float* __restrict__ a_scale,
                                                            const float* __restrict__ a_shift,
                                                            const float* __restrict__ G, int g_pitch, int g_col,
                                                            float* __restrict__ part, float* __restrict__ db_part) {
    __shared__ __align__(16) float As[BM * 64];
    __shared__ __align__(16) float Gs[BM * 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int q = tid & 15, rb = tid >> 4;
    const int ci_h = w >> 1, co_h = w & 1;
    const int ntiles = (g.rows_total + BM - 1) / BM;
    // grid.y walks 64-column blocks of G, grid.z groups of NT taps; every (y, z) owns a slab of partials
    tap0 += NT * blockIdx.z;
    g_col += 64 * blockIdx.y;
    {
        const long slot = (long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x;
        part += slot * NT * 4096;
        if (db_part) db_part += slot * 64;
        if (blockIdx.z != 0) db_part = nullptr;
    }
    const RowDecoder dec(g);
    floatx16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = floatx16{};
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);

    RowInfo ri[8];
    float4 rg[8], ra[8];
    bool ok[8];
    // rows of a tile: operand row info + the tile's gradient rows (zero past the end), first valid tap in flight
    auto open_tile = [&](int tile, const TileInfo& ti) {
        const int tfirst = first_tap(ti.mask);
        auto set = [&](int e, int n, int y, int x) {
            const int r = tile * BM + rb + 16 * e;
            ri[e] = make_row(g, r, n, y, x);
            const bool rv = r < g.rows_total;
            const int op = rv ? (n * g.Ho + y * g.oys + g.oy0) * g.Wo + x * g.oxs + g.ox0 : 0;
            float4 v = *reinterpret_cast<const float4*>(G + (long)op * g_pitch + g_col + 4 * q);
            v.x = rv ? v.x : 0.f; v.y = rv ? v.y : 0.f; v.z = rv ? v.z : 0.f; v.w = rv ? v.w : 0.f;
            rg[e] = v;
        };
        if (PM) dec.decode_pm<8>(g.pm_nimg, tile * BM + rb, 16, set);
        else dec.decode<8>(tile * BM + rb, 16, set);
#pragma unroll
        for (int e = 0; e < 8; ++e) ra[e] = load_a(g, A, ri[e], tap0 + tfirst, q, &ok[e]);
    };
    TileInfo ti = tile_info<PM>(g, (int)blockIdx.x * BM, BM, tap0, NT);
    unsigned mask = ti.mask;
    open_tile(blockIdx.x, ti);
    bool g_pending = true;                 // uniform: the tile's gradient rows still sit in registers
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        ti = tile_info<PM>(g, (tile + (int)gridDim.x) * BM, BM, tap0, NT);
        const unsigned nmask = ti.mask;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!((mask >> t) & 1u)) continue;                      // uniform: tap outside the image for the whole tile
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (HAS_T) {
                sc = *reinterpret_cast<const float4*>(a_scale + g.a_col[tap0 + t] + 4 * q);
                sh = *reinterpret_cast<const float4*>(a_shift + g.a_col[tap0 + t] + 4 * q);
            }
            __syncthreads();
            if (g_pending) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    *reinterpret_cast<float4*>(&Gs[(rb + 16 * e) * 64 + 4 * q]) = rg[e];
                    dbs.x += rg[e].x; dbs.y += rg[e].y; dbs.z += rg[e].z; dbs.w += rg[e].w;
                }
                g_pending = false;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *reinterpret_cast<float4*>(&As[(rb + 16 * e) * 64 + 4 * q]) = transform_a<HAS_T>(ra[e], ok[e], sc, sh);
            __syncthreads();
            const unsigned rest = taps_above(mask, t);
            if (rest) {
                const int tn = first_tap(rest);
#pragma unroll
                for (int e = 0; e < 8; ++e) ra[e] = load_a(g, A, ri[e], tap0 + tn, q, &ok[e]);
            } else {
                open_tile(tile + gridDim.x, ti);                   // past the last tile every row is invalid: element 0
                g_pending = true;
            }
            __builtin_amdgcn_sched_barrier(0);
            const float* ap = &As[h * 64 + 32 * ci_h + i];
            const float* gp = &Gs[h * 64 + 32 * co_h + i];
            float a0 = ap[0], a1 = ap[128], a2 = ap[256], a3 = ap[384];
            float g0 = gp[0], g1 = gp[128], g2 = gp[256], g3 = gp[384];
#pragma unroll
            for (int b4 = 0; b4 < BM / 8; ++b4) {
                float n0 = a0, n1 = a1, n2 = a2, n3 = a3, m0 = g0, m1 = g1, m2 = g2, m3 = g3;
                if (b4 + 1 < BM / 8) {
                    const float* an = ap + (b4 + 1) * 512;
                    const float* gn = gp + (b4 + 1) * 512;
                    n0 = an[0]; n1 = an[128]; n2 = an[256]; n3 = an[384];
                    m0 = gn[0]; m1 = gn[128]; m2 = gn[256]; m3 = gn[384];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, g0, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, g1, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, g2, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, g3, acc[t], 0, 0, 0);
                a0 = n0; a1 = n1; a2 = n2; a3 = n3; g0 = m0; g1 = m1; g2 = m2; g3 = m3;
            }
        }
        mask = nmask;
    }
    float* pb = part + (long)blockIdx.x * NT * 4096;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int ci = 32 * ci_h + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            pb[t * 4096 + ci * 64 + 32 * co_h + i] = acc[t][reg];
        }
    if (db_part) {
        // column sums of G: thread (rb, q) holds cols 4q..4q+3 over its rows
        __syncthreads();
        float* red = As;
        *reinterpret_cast<float4*>(&red[rb * 64 + 4 * q]) = dbs;
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) s += red[r * 64 + tid];
            db_part[(long)blockIdx.x * 64 + tid] = s;
        }
    }
}

// ---------------------------------------------------------------------------- weight gradient, streaming variant
// For launches whose image count is a multiple of 16.  Nothing is staged through LDS and the main loop has no
// barrier: the contraction runs over rows, and v_mfma_f32_16x16x4_f32 takes A[m][k] / B[k][n] with k = lane >> 4, so
// ONE 16-byte load per lane of an operand row (lane (i, kq): channels 4i..4i+3 of row kq of a 4-row group) and one of
// the gradient row are the fragments of 16 MFMAs: component ja of the operand against component jb of the gradient
// accumulates the (ci = 4m + ja, co = 4n + jb) sub-lattice of the 64x64 tap block.  A wave owns TWO taps (128
// accumulators) and walks a contiguous range of chunks (16 images at one grid position, positions fastest, so the
// shifted operand rows of neighbouring positions are re-read while still in L2); a tap outside the image at that
// position is skipped for the whole chunk, whatever the launch's row order.  Four 4-row groups = one chunk are in
// flight per wave (loads of chunk c+1 are issued as chunk c's groups are consumed).  The waves of a workgroup split
// into tap pairs x row streams and add their accumulators through LDS once, at the end.
typedef float floatx4 __attribute__((ext_vector_type(4)));
#if CMVAE_ABLATE == 20
__device__ long long g_clock_probe[2];
#endif

struct ChunkDesc {            // all uniform (SGPRs): the per-lane part of every address is a kernel-wide constant
    int a[2];                 // element offsets of the two taps' operand rows (image 16*ib, group 0)
    int go;                   // ... of the gradient row
    int sa[2], sg;            // offset step between 4-row groups (0 for a skipped tap / dead chunk)
    unsigned valid;           // bit u = tap u inside the image, bit 2 = chunk inside the range
};

// bid / nblk: the workgroup's index among those that share this geometry, taps and gradient columns; part / db_part
// point at that group's first partial (workgroup bid writes nt_stride tap blocks from part + bid * nt_stride * 4096).
template <bool HAS_T>
__device__ __forceinline__ void wgrad_stream_body(const ConvGeom& g, int tap0, int nt, int nimg,
                                                  const float* __restrict__ A, const float* __restrict__ a_scale,
                                                  const float* __restrict__ a_shift, const float* __restrict__ G,
                                                  int g_pitch, int g_col, float* __restrict__ part,
                                                  float* __restrict__ db_part, int bid, int nblk, int nt_stride,
                                                  float* comb) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const bool split = nt > 2;                                   // two tap pairs x two row streams, else four row streams
    const int tp = split ? (w & 1) : 0;
    const int p = split ? (w >> 1) : w;
    const int P = split ? 2 : 4;
    const int npix = g.Hg * g.Wg;
    const int nchunks = (nimg / 16) * npix;
    const int nstreams = nblk * P, sid = bid * P + p;
    const int c_begin = (int)((long)nchunks * sid / nstreams), c_end = (int)((long)nchunks * (sid + 1) / nstreams);
    // the wave's taps within the launch group.  Four taps are a product set {a,b} x {c,d} of row and column offsets:
    // pairing (a,c) with (b,d) and (a,d) with (b,c) gives both pairs the same number of in-image positions (85 / 84 of
    // a 7x7 grid; consecutive pairs would be 91 / 78 and the kernel would run at the pace of the heavier pair)
    const int t_u[2] = {split ? tp : 0, split ? 3 - tp : 1};
    const int img_a = g.Hi * g.Wi * g.a_pitch, img_g = g.Ho * g.Wo * g_pitch;

    float4 sc[2], sh[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        sc[u] = make_float4(1.f, 1.f, 1.f, 1.f); sh[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (HAS_T && t_u[u] < nt) {
            // times 2^-60: the fold is one clamped v_pk_fma_f32 per channel pair (fold_a_scaled); the accumulators
            // get the 2^60 back when they are written out
            sc[u] = scaled4(*reinterpret_cast<const float4*>(a_scale + g.a_col[tap0 + t_u[u]] + 4 * i), FOLD_DOWN);
            sh[u] = scaled4(*reinterpret_cast<const float4*>(a_shift + g.a_col[tap0 + t_u[u]] + 4 * i), FOLD_DOWN);
        }
    }
    // lane (i, kq): channels 4i .. 4i+3 of image kq of a 4-row group; everything else of an address is uniform, so the
    // loads take a scalar base and this one VGPR offset (no per-load vector address arithmetic)
    const int lane_a = kq * img_a + 4 * i, lane_g = kq * img_g + 4 * i;
    auto describe = [&](int ib, int y, int x, bool live) {
        ChunkDesc d;
        const int img = ib * 16;
        const int ys = y * g.sy, xs = x * g.sx;
        d.valid = live ? 4u : 0u;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bool in = false;
            int off = 0;
            if (t_u[u] < nt) {
                const int t = tap0 + t_u[u];
                const int dy = g.dy[t], dx = g.dx[t];
                in = live & ((unsigned)(ys + dy) < (unsigned)g.Hi) & ((unsigned)(xs + dx) < (unsigned)g.Wi);
                off = ((img * g.Hi + ys + dy) * g.Wi + xs + dx) * g.a_pitch + g.a_col[t];
            }
            d.a[u] = in ? off : 0;
            d.sa[u] = in ? 4 * img_a : 0;
            d.valid |= in ? (1u << u) : 0u;
        }
        const int op = (img * g.Ho + y * g.oys + g.oy0) * g.Wo + x * g.oxs + g.ox0;
        d.go = live ? op * g_pitch + g_col : 0;
        d.sg = live ? 4 * img_g : 0;
#if CMVAE_ABLATE == 7 || CMVAE_ABLATE == 8 || CMVAE_ABLATE == 9      // (8: and no tap is skipped) every operand load hits the same few rows: the MFMA-side ceiling of this loop
        d.a[0] = d.a[1] = d.go = 0; d.sa[0] = d.sa[1] = d.sg = 0;
#endif
        return d;
    };

    floatx4 acc[2][16];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[u][k] = floatx4{0.f, 0.f, 0.f, 0.f};
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
    const float db_on = (db_part != nullptr && tp == 0) ? 1.f : 0.f;

    float4 ra[4][2], rg[4];
    auto issue = [&](int slot, const ChunkDesc& d, int s) {
        const float* a0 = A + (d.a[0] + s * d.sa[0]);            // uniform pointers
        const float* a1 = A + (d.a[1] + s * d.sa[1]);
        const float* gp = G + (d.go + s * d.sg);
        ra[slot][0] = *reinterpret_cast<const float4*>(a0 + lane_a);
        ra[slot][1] = *reinterpret_cast<const float4*>(a1 + lane_a);
        rg[slot] = *reinterpret_cast<const float4*>(gp + lane_g);
    };
    auto consume = [&](int slot, unsigned valid) {
        const float4 gv = rg[slot];
        // every consumed chunk is inside the range (only the prefetch past the end is not); the bias-gradient sum is
        // two packed FMAs with a 0/1 factor instead of a branch or selects
        {
            floatx2 lo = {dbs.x, dbs.y}, hi = {dbs.z, dbs.w};
            lo = __builtin_elementwise_fma(floatx2{gv.x, gv.y}, floatx2{db_on, db_on}, lo);
            hi = __builtin_elementwise_fma(floatx2{gv.z, gv.w}, floatx2{db_on, db_on}, hi);
            dbs = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
        const float gb[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#if CMVAE_ABLATE != 8
            if (!((valid >> u) & 1u)) continue;                  // uniform
#endif
            const float4 a = fold_a_scaled_guarded<HAS_T>(ra[slot][u], sc[u], sh[u]);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int ja = 0; ja < 4; ++ja)
#pragma unroll
                for (int jb = 0; jb < 4; ++jb)
                    acc[u][ja * 4 + jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ja], gb[jb], acc[u][ja * 4 + jb], 0, 0, 0);
        }
    };

    int c = c_begin;
    int ib = c / npix;
    int y, x;
    { const int pix = c - ib * npix; y = pix / g.Wg; x = pix - y * g.Wg; }
#if CMVAE_ABLATE == 20      // shader clock during this kernel: clock64 ticks per 100 MHz wall tick, seen by one wave
    const long long clk0 = clock64(), wall0 = wall_clock64();
#endif
#if CMVAE_STAGGER
    // the two workgroups of a CU start half a chunk apart, so that one's address / descriptor work falls into the
    // other's MFMA phase instead of both alternating in lock step
    if (bid >= nblk / 2) __builtin_amdgcn_s_sleep(CMVAE_STAGGER);
#endif
    ChunkDesc cur = describe(ib, y, x, c < c_end);
#pragma unroll
    for (int s = 0; s < 4; ++s) issue(s, cur, s);
    for (; c < c_end; ++c) {
        if (++x == g.Wg) { x = 0; if (++y == g.Hg) { y = 0; ++ib; } }
        const ChunkDesc nxt = describe(ib, y, x, c + 1 < c_end);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            consume(s, cur.valid);
#if CMVAE_ABLATE != 9          // 9: no loads in the main loop at all (operands stay what the prologue fetched)
            issue(s, nxt, s);
#endif
        }
        cur = nxt;
    }

#if CMVAE_ABLATE == 20
    if (bid == 7 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) {
        g_clock_probe[0] = clock64() - clk0; g_clock_probe[1] = wall_clock64() - wall0;
    }
#endif
    if (HAS_T) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[u][k] *= FOLD_UP;          // exact: the operand was folded times 2^-60
    }
    // add the row streams of a tap pair (wave p == 0 keeps the sum): lane-major LDS image of the 128 accumulators
    float* cb = comb + tp * 8192 + lane;
    for (int k = 1; k < P; ++k) {
        if (p == k) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 64; ++r) cb[(u * 64 + r) * 64] = acc[u][r >> 2][r & 3];
        }
        __syncthreads();
        if (p == 0) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 64; ++r) acc[u][r >> 2][r & 3] += cb[(u * 64 + r) * 64];
        }
        __syncthreads();
    }
    if (db_part) {
        dbs.x += __shfl_xor(dbs.x, 16); dbs.y += __shfl_xor(dbs.y, 16); dbs.z += __shfl_xor(dbs.z, 16); dbs.w += __shfl_xor(dbs.w, 16);
        dbs.x += __shfl_xor(dbs.x, 32); dbs.y += __shfl_xor(dbs.y, 32); dbs.z += __shfl_xor(dbs.z, 32); dbs.w += __shfl_xor(dbs.w, 32);
        if (tp == 0 && kq == 0) *reinterpret_cast<float4*>(&comb[p * 64 + 4 * i]) = dbs;
        __syncthreads();
        if (tid < 64) {
            float sum = 0.f;
            for (int k = 0; k < P; ++k) sum += comb[k * 64 + tid];
            db_part[(long)bid * 64 + tid] = sum;
        }
    }
    if (p == 0) {
        float* pb = part + (long)bid * nt_stride * 4096;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (t_u[u] >= nt) continue;
#pragma unroll
            for (int ja = 0; ja < 4; ++ja)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = 16 * kq + 4 * r + ja;
                    *reinterpret_cast<float4*>(&pb[t_u[u] * 4096 + ci * 64 + 4 * i]) =
                        make_float4(acc[u][ja * 4 + 0][r], acc[u][ja * 4 + 1][r], acc[u][ja * 4 + 2][r], acc[u][ja * 4 + 3][r]);
                }
        }
    }
}

// ---------------------------------------------------------------------------- weight gradient on the bf16 pipe
// The streaming kernel above with its products on the bf16 matrix pipe (mfma_split6's scheme, here with
// v_mfma_f32_16x16x32_bf16: k = 8 (lane >> 4) + e, e = 0..7), for the four-tap phases of the 4 x 4 transposed convs.
// A chunk is 32 images at one grid position; lane (i, kg) loads channels 4i .. 4i+3 of images 8 kg .. 8 kg + 7 (eight
// 16-byte loads per operand, each instruction 4 x 256 contiguous bytes), so component ja of its eight registers IS the
// A fragment of sub-lattice ja and component jb of the gradient's the B fragment of jb — split into three bf16 planes
// in registers, no transposition anywhere.  The accumulator layout, the combination of the row streams and the
// partials are those of wgrad_stream_body.  One workgroup per CU (512 registers per lane: 128 accumulators, the raw
// rows of the next chunk in flight, 3 x 48 fragment registers): the split of one fragment is interleaved with the MFMAs
// of the previous one.  (Phases with 1 / 2 taps — the 3 x 3 transposed conv — stay on the fp32 streaming kernel: a
// general two-taps-per-wave version of this kernel ran layer 2 in 350 us against 298.)
__device__ __forceinline__ floatx4 mfma16_split6(const Frag3& a, const Frag3& b, floatx4 c) {       // small terms first
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[2]), as_bf(b.p[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[0]), as_bf(b.p[2]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[1]), as_bf(b.p[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[1]), as_bf(b.p[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[0]), as_bf(b.p[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf(a.p[0]), as_bf(b.p[0]), c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ float comp4(const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; }
// component j of eight rows -> the three planes of one 8-value fragment
__device__ __forceinline__ Frag3 split_rows(const float4 (&r)[8], int j, const SplitSel& q) {
    const Split3 a = split_pair(comp4(r[0], j), comp4(r[1], j), q), b = split_pair(comp4(r[2], j), comp4(r[3], j), q),
                 c = split_pair(comp4(r[4], j), comp4(r[5], j), q), d = split_pair(comp4(r[6], j), comp4(r[7], j), q);
    Frag3 f;
    f.p[0] = make_uint4(a.p1, b.p1, c.p1, d.p1);
    f.p[1] = make_uint4(a.p2, b.p2, c.p2, d.p2);
    f.p[2] = make_uint4(a.p3, b.p3, c.p3, d.p3);
    return f;
}

// Four-tap phases {dyA, dyB} x {0, dxr} (the 4 x 4 transposed convs): a wave's two taps sit in ONE operand row, at
// columns x and x + dxr.  Walking the positions of a grid row against dxr, the rows read for the unshifted tap at one
// position ARE the shifted tap's rows at the next: they stay in registers as split fragments, so a chunk requests and
// splits one tap's operand rows instead of two (half of the operand requests and of the split's VALU work).  At the start of a grid row the shifted tap is outside the image (zero fragments);
// where it is not — the first chunk of a stream that starts inside a row — its rows are fetched for that chunk alone.
template <bool HAS_T>
__device__ __forceinline__ void wgrad_bf3_row_body(const ConvGeom& g, int nimg, const float* __restrict__ A,
                                                   const float* __restrict__ a_scale, const float* __restrict__ a_shift,
                                                   const float* __restrict__ G, int g_pitch, float* __restrict__ part,
                                                   float* __restrict__ db_part, int bid, int nblk, float* comb) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const SplitSel ssel = make_split_sel();
    const int tp = w & 1, p = w >> 1;                            // tap pair (operand row dyA / dyB) x row stream
    constexpr int P = 2;
    const int npix = g.Hg * g.Wg;
    const int nchunks = (nimg / 32) * npix;
    const int nstreams = nblk * P, sid = bid * P + p;
    const int c_begin = (int)((long)nchunks * sid / nstreams), c_end = (int)((long)nchunks * (sid + 1) / nstreams);
    // taps 2 tp and 2 tp + 1 share dy; un: the one with dx = 0 (its rows are requested), ur: the shifted one (reused)
    const int un = g.dx[2 * tp] == 0 ? 0 : 1, ur = 1 - un;
    const int tn = 2 * tp + un, tr = 2 * tp + ur;
    const int dy = g.dy[tn], dxr = g.dx[tr];
    const bool ascending = dxr < 0;                              // walk against the shift
    const int img_a = g.Hi * g.Wi * g.a_pitch, img_g = g.Ho * g.Wo * g_pitch;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (HAS_T) {
        sc = scaled4(*reinterpret_cast<const float4*>(a_scale + g.a_col[tn] + 4 * i), FOLD_DOWN);
        sh = scaled4(*reinterpret_cast<const float4*>(a_shift + g.a_col[tn] + 4 * i), FOLD_DOWN);
    }
    // Requests are buffer loads: the lane's byte offset of each of its eight images is a kernel-wide constant in a
    // register (voffset), the chunk's offset a scalar (soffset), the tensor a descriptor — no address arithmetic per
    // request (as global loads each took ~7 scalar instructions or a 64-bit vector add); reads past the tensor's end
    // would return zero.  (The caller checks that both tensors are smaller than 4 GB.)
    typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (unsigned)((long)nimg * img_a * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G), 0, (unsigned)((long)nimg * img_g * 4), 0x00020000);
    unsigned voff_a[8], voff_g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        voff_a[j] = (unsigned)((8 * kq + j) * img_a + 4 * i) * 4u;
        voff_g[j] = (unsigned)((8 * kq + j) * img_g + 4 * i) * 4u;
    }
    auto load16 = [](const __amdgpu_buffer_rsrc_t& r, unsigned voff, int soff_elems) {
        const uintx4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff_elems * 4, 0);
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    };
    struct Desc { int a, go; unsigned flags; };                      // flags: 1 = row in image, 2 = shifted tap in image,
                                                                 //        4 = first position of a grid row walk, 8 = live
    auto describe = [&](int c, bool live) {
        Desc d;
        const int ib = c / npix, pix = c - ib * npix;
        const int y = pix / g.Wg, xi = pix - y * g.Wg;
        const int x = ascending ? xi : g.Wg - 1 - xi;
        const int img = ib * 32, ys = y * g.sy + dy, xs = x * g.sx;
        const bool row_in = live & ((unsigned)ys < (unsigned)g.Hi) & ((unsigned)xs < (unsigned)g.Wi);
        const bool sh_in = row_in & ((unsigned)(xs + dxr) < (unsigned)g.Wi);
        d.a = row_in ? ((img * g.Hi + ys) * g.Wi + xs) * g.a_pitch + g.a_col[tn] : 0;
        const int op = (img * g.Ho + y * g.oys + g.oy0) * g.Wo + x * g.oxs + g.ox0;
        d.go = live ? op * g_pitch : 0;
        d.flags = (row_in ? 1u : 0u) | (sh_in ? 2u : 0u) | (xi == 0 ? 4u : 0u) | (live ? 8u : 0u);
#if CMVAE_ABLATE == 7 || CMVAE_ABLATE == 74      // every load hits the same few rows: the compute-side ceiling of this loop
        d.a = d.go = 0;
#elif CMVAE_ABLATE == 71   // ... only the operand's loads do
        d.a = 0;
#elif CMVAE_ABLATE == 72   // ... only the gradient's
        d.go = 0;
#endif
        return d;
    };

    floatx4 acc[2][16];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[u][k] = floatx4{0.f, 0.f, 0.f, 0.f};
    float4 dbs = make_float4(0.f, 0.f, 0.f, 0.f);
    const float db_on = (db_part != nullptr && tp == 0) ? 1.f : 0.f;

    float4 rx[8], rg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rx[j] = make_float4(0.f, 0.f, 0.f, 0.f);      // (finite, whatever is skipped first)
    // Rows outside the image (and the prefetch past the end of the range) are not requested (uniform branches; the
    // registers' stale, finite contents are multiplied by zero).
    auto issue_x = [&](const Desc& d) {
        if (!(d.flags & 1u)) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) rx[j] = load16(rsrc_a, voff_a[j], d.a);
    };
    auto issue_g = [&](const Desc& d) {
        if (!(d.flags & 8u)) return;
#pragma unroll
        for (int j = 0; j < 8; ++j) rg[j] = load16(rsrc_g, voff_g[j], d.go);
    };
    auto folded = [&](float4 (&a)[8], float von) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 f = fold_a_scaled<HAS_T>(rx[j], sc, sh);
            const floatx2 lo = floatx2{f.x, f.y} * floatx2{von, von}, hi = floatx2{f.z, f.w} * floatx2{von, von};
            a[j] = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
    };

    // A fragments of the unshifted tap: chunk k writes xf[k & 1], the shifted tap of chunk k reads xf[(k & 1) ^ 1] (the
    // loop is unrolled by two, so nothing is copied from "new" to "previous")
    Frag3 xf[2][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ja = 0; ja < 4; ++ja) xf[q][ja].p[0] = xf[q][ja].p[1] = xf[q][ja].p[2] = make_uint4(0u, 0u, 0u, 0u);
    int c = c_begin;
    Desc cur = describe(c, c < c_end);
    if (c < c_end && (cur.flags & 2u) && !(cur.flags & 4u)) {
        // the stream starts inside a grid row: the shifted tap's rows of this first chunk (= the unshifted tap's of the
        // position before it in the walk)
        Desc pre = cur;
        pre.a = cur.a + dxr * g.a_pitch;
#if CMVAE_ABLATE == 7 || CMVAE_ABLATE == 71 || CMVAE_ABLATE == 74      // (the ablation builds zero cur.a: stay inside the tensor)
        pre.a = 0;
#endif
        issue_x(pre);
        float4 a[8];
        folded(a, 1.f);
#pragma unroll
        for (int ja = 0; ja < 4; ++ja) xf[1][ja] = split_rows(a, ja, ssel);      // chunk 0 reads xf[1] as "previous"
    }
    issue_x(cur);
    issue_g(cur);
#if CMVAE_ABLATE == 73 || CMVAE_ABLATE == 74     // how long does a wave wait for the rows it requested one chunk earlier (10 ns
                                                // ticks), and at what shader clock does the loop run?  (74: every load a hit;
                                                // tools/probe_wgrad_wait.py)
    unsigned long long wait_total = 0;
    const unsigned long long loop_t0 = wall_clock64();
    const long long loop_c0 = clock64();
#endif
    auto chunk = [&](Frag3 (&xnew)[4], Frag3 (&xprev)[4], const Desc& cur, const Desc& nxt) {
#if CMVAE_ABLATE == 73 || CMVAE_ABLATE == 74
        {
            const unsigned long long tw0 = wall_clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wait_total += wall_clock64() - tw0;
        }
#endif
        float4 a[8];
        folded(a, (cur.flags & 1u) ? 1.f : 0.f);
        __builtin_amdgcn_sched_barrier(0);
        issue_x(nxt);
        __builtin_amdgcn_sched_barrier(0);
        {
            floatx2 lo = {dbs.x, dbs.y}, hi = {dbs.z, dbs.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lo = __builtin_elementwise_fma(floatx2{rg[j].x, rg[j].y}, floatx2{db_on, db_on}, lo);
                hi = __builtin_elementwise_fma(floatx2{rg[j].z, rg[j].w}, floatx2{db_on, db_on}, hi);
            }
            dbs = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
        Frag3 gB[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) gB[jb] = split_rows(rg, jb, ssel);
        __builtin_amdgcn_sched_barrier(0);
        issue_g(nxt);
        __builtin_amdgcn_sched_barrier(0);
        if (!(cur.flags & 2u)) {                                 // uniform: the shifted tap is outside the image here
#pragma unroll
            for (int ja = 0; ja < 4; ++ja) xprev[ja].p[0] = xprev[ja].p[1] = xprev[ja].p[2] = make_uint4(0u, 0u, 0u, 0u);
        }
        xnew[0] = split_rows(a, 0, ssel);
        __builtin_amdgcn_sched_barrier(0);
        // stages 0..3: the unshifted tap, the next sub-lattice's fragment split in the gaps of the 24 MFMAs;
        // stages 4..7: the shifted tap on the fragments kept from the previous position
#pragma unroll
        for (int ja = 0; ja < 4; ++ja) {
            if (ja < 3) xnew[ja + 1] = split_rows(a, ja + 1, ssel);
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[0][ja * 4 + jb] = mfma16_split6(xnew[ja], gB[jb], acc[0][ja * 4 + jb]);
            if (ja < 3) {
#pragma unroll
                for (int m = 0; m < 14; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 10, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ja = 0; ja < 4; ++ja) {
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) acc[1][ja * 4 + jb] = mfma16_split6(xprev[ja], gB[jb], acc[1][ja * 4 + jb]);
        }
    };
    for (; c < c_end; c += 2) {
        const Desc nxt = describe(c + 1, c + 1 < c_end);
        chunk(xf[0], xf[1], cur, nxt);
        if (c + 1 >= c_end) break;                               // uniform
        cur = describe(c + 2, c + 2 < c_end);
        chunk(xf[1], xf[0], nxt, cur);
    }

#if CMVAE_ABLATE == 73 || CMVAE_ABLATE == 74
    if (lane == 0 && nchunks > 20000 && bid < 1024) {           // (layer 3 only) per wave: total wait, shader cycles, loop time, chunks
        unsigned long long* o = g_wg_times + (bid * 4 + w) * 3;
        o[0] = wait_total; o[1] = (unsigned long long)(clock64() - loop_c0);
        o[2] = ((wall_clock64() - loop_t0) << 20) | (unsigned long long)(c_end - c_begin);
    }
#endif
    if (HAS_T) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[u][k] *= FOLD_UP;          // exact: the operand was folded times 2^-60
    }
    float* cb = comb + tp * 8192 + lane;
    if (p == 1) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 64; ++r) cb[(u * 64 + r) * 64] = acc[u][r >> 2][r & 3];
    }
    __syncthreads();
    if (p == 0) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 64; ++r) acc[u][r >> 2][r & 3] += cb[(u * 64 + r) * 64];
    }
    __syncthreads();
    if (db_part) {
        dbs.x += __shfl_xor(dbs.x, 16); dbs.y += __shfl_xor(dbs.y, 16); dbs.z += __shfl_xor(dbs.z, 16); dbs.w += __shfl_xor(dbs.w, 16);
        dbs.x += __shfl_xor(dbs.x, 32); dbs.y += __shfl_xor(dbs.y, 32); dbs.z += __shfl_xor(dbs.z, 32); dbs.w += __shfl_xor(dbs.w, 32);
        if (tp == 0 && kq == 0) *reinterpret_cast<float4*>(&comb[p * 64 + 4 * i]) = dbs;
        __syncthreads();
        if (tid < 64) db_part[(long)bid * 64 + tid] = comb[tid] + comb[64 + tid];
    }
    if (p == 0) {
        float* pb = part + (long)bid * 4 * 4096;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = u == 0 ? tn : tr;                      // acc[0]: the unshifted tap, acc[1]: the shifted one
#pragma unroll
            for (int ja = 0; ja < 4; ++ja)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ci = 16 * kq + 4 * r + ja;
                    *reinterpret_cast<float4*>(&pb[t * 4096 + ci * 64 + 4 * i]) =
                        make_float4(acc[u][ja * 4 + 0][r], acc[u][ja * 4 + 1][r], acc[u][ja * 4 + 2][r], acc[u][ja * 4 + 3][r]);
                }
        }
    }
}

// the sub-pixel phases of a transposed convolution in one launch (see conv_wgrad_stream_multi_kernel), bf16 pipe
// the sub-pixel phases of a transposed convolution in one launch (see conv_wgrad_stream_multi_kernel), bf16 pipe; every
// phase has four taps in two operand rows
template <bool HAS_T>
__global__ __launch_bounds__(256, 1) void conv_wgrad_bf3_multi_kernel(ConvGeomMulti p, int nimg, int db_blocks,
                                                                      const float* __restrict__ A,
                                                                      const float* __restrict__ a_scale,
                                                                      const float* __restrict__ a_shift,
                                                                      const float* __restrict__ G, int g_pitch,
                                                                      float* __restrict__ part,
                                                                      float* __restrict__ db_part) {
    __shared__ __align__(16) float comb[2 * 128 * 64];
    const int b = blockIdx.x;
    const int ph = (b >= p.begin[1]) + (b >= p.begin[2]) + (b >= p.begin[3]);          // uniform
    const ConvGeom& g = p.g[ph];
    const int b0 = p.begin[ph];
    wgrad_bf3_row_body<HAS_T>(g, nimg, A, a_scale, a_shift, G, g_pitch, part + (long)b0 * 4 * 4096,
                              (db_part && b < db_blocks) ? db_part + (long)b0 * 64 : nullptr, b - b0,
                              p.begin[ph + 1] - b0, comb);
}

template <bool HAS_T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_stream_kernel(ConvGeom g, int tap0, int nt, int nimg,
                                                                   const float* __restrict__ A,
                                                                   const float* __restrict__ a_scale,
                                                                   const float* __restrict__ a_shift,
                                                                   const float* __restrict__ G, int g_pitch, int g_col,
                                                                   float* __restrict__ part,
                                                                   float* __restrict__ db_part) {
    __shared__ __align__(16) float comb[2 * 128 * 64];
    const long slot = (long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x;
    wgrad_stream_body<HAS_T>(g, tap0 + nt * (int)blockIdx.z, nt, nimg, A, a_scale, a_shift, G, g_pitch,
                             g_col + 64 * (int)blockIdx.y, part + slot * nt * 4096,
                             (db_part && blockIdx.z == 0) ? db_part + slot * 64 : nullptr, (int)blockIdx.x,
                             (int)gridDim.x, nt, comb);
}

// The sub-pixel phases of a transposed convolution (ConvGeomMulti, see conv_gemm_wstat_multi_kernel) in one launch:
// workgroups begin[k] .. begin[k+1]-1 stream phase k's rows.  Every workgroup leaves 4 tap blocks of partials (a
// phase with fewer taps leaves the rest unwritten and unread) and one bias-gradient partial: every phase covers
// output pixels of its own, so the bias gradient is the sum over all of them.
// Workgroups from db_blocks on leave no bias-gradient partial (tap groups of ONE geometry share their gradient rows).
template <bool HAS_T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_stream_multi_kernel(ConvGeomMulti p, int nimg, int db_blocks,
                                                                         const float* __restrict__ A,
                                                                         const float* __restrict__ a_scale,
                                                                         const float* __restrict__ a_shift,
                                                                         const float* __restrict__ G, int g_pitch,
                                                                         float* __restrict__ part,
                                                                         float* __restrict__ db_part) {
    __shared__ __align__(16) float comb[2 * 128 * 64];
    const int b = blockIdx.x;
    const int ph = (b >= p.begin[1]) + (b >= p.begin[2]) + (b >= p.begin[3]);          // uniform
    const ConvGeom& g = p.g[ph];
    const int b0 = p.begin[ph];
    wgrad_stream_body<HAS_T>(g, 0, g.ntaps, nimg, A, a_scale, a_shift, G, g_pitch, 0, part + (long)b0 * 4 * 4096,
                             (db_part && b < db_blocks) ? db_part + (long)b0 * 64 : nullptr, b - b0,
                             p.begin[ph + 1] - b0, 4, comb);
}

// out[y*y_stride + tap_off[NT*z + t] + ci*s_ci + co*s_co] = sum_blk part[(z,y)][blk][t][ci][co];
// db[64*y + co] = sum_blk db_part[(0,y)][blk][co]
struct WgradOut {
    long tap_off[32];
    long s_ci, s_co, y_stride;
};
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ part, int nblk, int NT,
                                                            WgradOut o, float* __restrict__ dW,
                                                            const float* __restrict__ db_part,
                                                            float* __restrict__ db, int accumulate_db) {
    // 64 consecutive elements x 16 groups of partial blocks per workgroup (enough loads in flight to cover the
    // latency of up to 512 partials per element); fixed summation order
    __shared__ float red[16 * 64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + c;
    const long slot = (long)(blockIdx.z * gridDim.y + blockIdx.y) * nblk;
    part += slot * NT * 4096;
    const bool is_db = blockIdx.x == NT * 64;                  // extra workgroup: the bias gradient
    if (is_db && (db == nullptr || blockIdx.z != 0)) return;
    float s0 = 0.f, s1 = 0.f;
    if (!is_db) {
        int b = g;
        for (; b + 112 < nblk; b += 128) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = part[(long)(b + 16 * i) * NT * 4096 + e];
#pragma unroll
            for (int i = 0; i < 8; i += 2) { s0 += v[i]; s1 += v[i + 1]; }
        }
        for (; b + 16 < nblk; b += 32) {
            s0 += part[(long)b * NT * 4096 + e];
            s1 += part[(long)(b + 16) * NT * 4096 + e];
        }
        if (b < nblk) s0 += part[(long)b * NT * 4096 + e];
    } else {
        db_part += slot * 64;
        for (int b = g; b < nblk; b += 16) s0 += db_part[(long)b * 64 + c];
    }
    red[g * 64 + c] = s0 + s1;
    __syncthreads();
    if (g != 0) return;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r * 64 + c];
    if (!is_db) {
        const int t = e >> 12, ci = (e >> 6) & 63, co = e & 63;
        dW[blockIdx.y * o.y_stride + o.tap_off[NT * blockIdx.z + t] + ci * o.s_ci + co * o.s_co] = s;
    } else {
        float* d = db + 64 * blockIdx.y;
        d[c] = accumulate_db ? d[c] + s : s;
    }
}

// The same for a multi-phase launch: phase y's partials are those of workgroups begin[y] .. begin[y+1]-1 (4 tap blocks
// each), its taps go to tap_off[4*y + t]; the bias gradient sums every workgroup's partial.
struct WgradMultiOut {
    long tap_off[16];
    long s_ci, s_co;
    int begin[5], ntaps[4];
};
__global__ __launch_bounds__(1024) void wgrad_reduce_multi_kernel(const float* __restrict__ part, WgradMultiOut o,
                                                                  float* __restrict__ dW,
                                                                  const float* __restrict__ db_part,
                                                                  float* __restrict__ db, int nblk_all,
                                                                  int accumulate_db) {
    __shared__ float red[16 * 64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + c, ph = blockIdx.y;
    const bool is_db = blockIdx.x == 4 * 64;
    if (is_db ? (db == nullptr || ph != 0) : (e >> 12) >= o.ntaps[ph]) return;      // uniform per workgroup
    float s0 = 0.f, s1 = 0.f;
    if (!is_db) {
        const int nblk = o.begin[ph + 1] - o.begin[ph];
        const float* pp = part + (long)o.begin[ph] * 4 * 4096 + e;
        int b = g;
        for (; b + 112 < nblk; b += 128) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = pp[(long)(b + 16 * i) * 4 * 4096];
#pragma unroll
            for (int i = 0; i < 8; i += 2) { s0 += v[i]; s1 += v[i + 1]; }
        }
        for (; b + 16 < nblk; b += 32) {
            s0 += pp[(long)b * 4 * 4096];
            s1 += pp[(long)(b + 16) * 4 * 4096];
        }
        if (b < nblk) s0 += pp[(long)b * 4 * 4096];
    } else {
        for (int b = g; b < nblk_all; b += 16) s0 += db_part[(long)b * 64 + c];
    }
    red[g * 64 + c] = s0 + s1;
    __syncthreads();
    if (g != 0) return;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += red[r * 64 + c];
    if (!is_db) {
        const int t = e >> 12, ci = (e >> 6) & 63, co = e & 63;
        dW[o.tap_off[4 * ph + t] + ci * o.s_ci + co * o.s_co] = s;
    } else {
        db[c] = accumulate_db ? db[c] + s : s;
    }
}

static int check_geom(const ConvGeom& g) {
    CMVAE_REQUIRE(g.rows_total >= 0 && g.Hg > 0 && g.Wg > 0 && g.Hi > 0 && g.Wi > 0 && g.Ho > 0 && g.Wo > 0,
                  "conv geometry: non-positive extent");
    CMVAE_REQUIRE(g.ntaps >= 1 && g.ntaps <= 32, "conv geometry: ntaps=%d outside [1,32]", g.ntaps);
    CMVAE_REQUIRE((long)g.Hg * g.Wg <= (1L << 20), "conv geometry: per-image grid larger than 2^20 rows");
    CMVAE_REQUIRE(g.a_pitch >= 64 && g.a_pitch % 4 == 0 && g.o_pitch >= 64, "conv geometry: bad pitch");
    CMVAE_REQUIRE(g.rows_total % (g.Hg * g.Wg) == 0, "conv geometry: rows not a multiple of the per-image grid");
    CMVAE_REQUIRE((g.Hg - 1) * g.oys + g.oy0 < g.Ho && (g.Wg - 1) * g.oxs + g.ox0 < g.Wo && g.oy0 >= 0 && g.ox0 >= 0,
                  "conv geometry: output grid exceeds the output image");
    CMVAE_REQUIRE((long)(g.rows_total / (g.Hg * g.Wg)) * g.Hi * g.Wi * g.a_pitch < (1L << 31),
                  "conv geometry: input larger than 2^31 elements");
    for (int t = 0; t < g.ntaps; ++t)
    {
        CMVAE_REQUIRE(g.a_col[t] >= 0 && g.a_col[t] + 64 <= g.a_pitch && g.a_col[t] % 4 == 0,
                      "conv geometry: tap %d column window outside the input row", t);
        CMVAE_REQUIRE(g.w_off[t] >= 0 && g.w_off[t] % 4 == 0, "conv geometry: tap %d weight offset", t);
    }
    return 0;
}

}  // namespace cmvae

using namespace cmvae;

static ConvGeom to_geom(const cmvae_conv_geom* c) {
    ConvGeom g{};
    g.rows_total = c->rows_total; g.Hg = c->Hg; g.Wg = c->Wg; g.Hi = c->Hi; g.Wi = c->Wi; g.sy = c->sy; g.sx = c->sx;
    g.Ho = c->Ho; g.Wo = c->Wo; g.oys = c->oys; g.oxs = c->oxs; g.oy0 = c->oy0; g.ox0 = c->ox0; g.ntaps = c->ntaps;
    g.a_pitch = c->a_pitch; g.o_pitch = c->o_pitch; g.w_pitch = c->w_pitch;
    g.w_nblk_stride = c->w_nblk_stride;
    // pixel-major only pays (and only keeps its tiles at one grid position) from a tile's worth of images on
    const int npix = c->Hg * c->Wg;
    g.pm_nimg = (c->pixel_major && npix > 0 && c->rows_total / npix >= 64) ? c->rows_total / npix : 0;
    g.xcd_order = 1;
    for (int t = 0; t < 32; ++t) { g.dy[t] = c->dy[t]; g.dx[t] = c->dx[t]; g.a_col[t] = c->a_col[t]; g.w_off[t] = c->w_off[t]; }
    return g;
}


static int gemm_grid(int rows_total, int nblk_cols) {
    const int ntiles = (rows_total + BM - 1) / BM;
    int gx = 3 * compute_units() / (nblk_cols > 0 ? nblk_cols : 1);      // launch bounds: three workgroups per CU
    if (gx < 64) gx = 64;
    return ntiles < gx ? (ntiles > 0 ? ntiles : 1) : gx;
}

// weight-gradient blocks: two per CU (64 KB of LDS each), every block leaves one partial per tap
static int wgrad_grid(int rows_total) {
    const int ntiles = (rows_total + BM - 1) / BM;
    const int cap = 2 * compute_units();
    return ntiles < cap ? (ntiles > 0 ? ntiles : 1) : cap;
}

extern "C" int cmvae_conv_gemm_grid(int rows_total, int n_col_blocks) {
    cmvae::clear_stale_error(); return gemm_grid(rows_total, n_col_blocks); }

extern "C" int cmvae_conv_bf16_pipe_min_rows(void) { cmvae::clear_stale_error(); return bf3_min_rows(); }

extern "C" int cmvae_conv_gemm_multi_grid(void) { cmvae::clear_stale_error(); return 2 * compute_units(); }

extern "C" int cmvae_conv_gemm_multi(const cmvae_conv_geom* geoms, int ngeom, const float* A, const float* Wt,
                                     const float* bias, const float* a_scale, const float* a_shift, float* out,
                                     float* stats_part, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(geoms && A && Wt && out, "cmvae_conv_gemm_multi: NULL argument");
    CMVAE_REQUIRE(ngeom >= 1 && ngeom <= 4, "cmvae_conv_gemm_multi: %d geometries (1..4)", ngeom);
    CMVAE_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "cmvae_conv_gemm_multi: scale and shift go together");
    ConvGeomMulti p{};
    long work[4] = {0, 0, 0, 0}, total_work = 0, total_tiles = 0;
    int max_taps = 0;
    // large pixel-major launches (the decoder's layers 2 and 3): products on the bf16 pipe, 256-row tiles, one
    // workgroup of 8 waves per CU
    long all_rows = 0;
    for (int k = 0; k < ngeom; ++k) all_rows += geoms[k].rows_total;
    const bool bf3 = to_geom(&geoms[0]).pm_nimg > 0 && all_rows >= (long)bf3_min_rows();
    const int tile_rows = bf3 ? 256 : 128;
    for (int k = 0; k < ngeom; ++k) {
        p.g[k] = to_geom(&geoms[k]);
        const ConvGeom& g = p.g[k];
        if (int e = check_geom(g)) return e;
        CMVAE_REQUIRE(g.ntaps <= WSTAT_TAPS4, "cmvae_conv_gemm_multi: geometry %d has %d taps (at most %d)", k, g.ntaps,
                      WSTAT_TAPS4);
        CMVAE_REQUIRE(g.rows_total > 0 && g.o_pitch >= 64, "cmvae_conv_gemm_multi: geometry %d is empty", k);
        CMVAE_REQUIRE((g.pm_nimg > 0) == (p.g[0].pm_nimg > 0), "cmvae_conv_gemm_multi: mixed row orders");
        const long tiles = (g.rows_total + tile_rows - 1) / tile_rows;
        work[k] = tiles * g.ntaps;
        total_work += work[k];
        total_tiles += tiles;
        if (g.ntaps > max_taps) max_taps = g.ntaps;
    }
    for (int k = ngeom; k < 4; ++k) p.g[k] = p.g[0];
    // workgroups per geometry in proportion to its tap work, at least one, at most its tiles
    int cap = (bf3 ? 1 : 2) * compute_units();
    if (total_tiles < cap) cap = (int)total_tiles;
    if (cap < ngeom) cap = ngeom;
    int nb[4] = {0, 0, 0, 0}, used = 0;
    for (int k = 0; k < ngeom; ++k) {
        long n = (long)cap * work[k] / total_work;
        const long tiles = (p.g[k].rows_total + tile_rows - 1) / tile_rows;
        if (n < 1) n = 1;
        if (n > tiles) n = tiles;
        nb[k] = (int)n;
        used += nb[k];
    }
    // pixel-major launches: multiples of 8 workgroups per geometry, so that every geometry's workgroups spread evenly
    // over the 8 XCDs (the tile order of wstat_body counts on it); what the rounding frees goes to the largest
    if (p.g[0].pm_nimg > 0 && p.g[0].xcd_order) {
        bool all8 = true;
        for (int k = 0; k < ngeom; ++k) all8 = all8 && nb[k] >= 8;
        if (all8) {
            int freed = 0;
            for (int k = 0; k < ngeom; ++k) { freed += nb[k] & 7; nb[k] &= ~7; }
            for (int pass = 0; freed >= 8 && pass < 64; ++pass) {
                int best = -1;
                for (int k = 0; k < ngeom; ++k) {
                    const long tiles = (p.g[k].rows_total + tile_rows - 1) / tile_rows;
                    if (nb[k] + 8 <= tiles && (best < 0 || work[k] * nb[best] > work[best] * nb[k])) best = k;
                }
                if (best < 0) break;
                nb[best] += 8; freed -= 8;
            }
            used = 0;
            for (int k = 0; k < ngeom; ++k) used += nb[k];
        }
    }
    p.begin[0] = 0;
    for (int k = 0; k < 4; ++k) p.begin[k + 1] = p.begin[k] + nb[k];
    // (begin[k] of an unused geometry is the grid size: never reached)
    const int stats_rows = 2 * compute_units();
    const int nw = bf3 ? 8 : 4;
    const size_t lds = (bf3 ? (size_t)max_taps * BF3_TAP_BYTES : (size_t)max_taps * 64 * LDA * sizeof(float)) +
                       2 * nw * 32 * sizeof(int) + nw * 128 * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    const bool pm = p.g[0].pm_nimg > 0;
#define CMVAE_MULTI_BF3(HAS_T)                                                                                         \
    do {                                                                                                               \
        static const bool opted = [] {                                                                                 \
            (void)hipFuncSetAttribute((const void*)conv_gemm_wstat_multi_kernel<0, HAS_T, true, true>,                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
            (void)hipGetLastError();                                                                                   \
            return true;                                                                                               \
        }();                                                                                                           \
        (void)opted;                                                                                                   \
        hipLaunchKernelGGL((conv_gemm_wstat_multi_kernel<0, HAS_T, true, true>), dim3(used), dim3(512), lds, st, p, A, \
                           Wt, bias, a_scale, a_shift, out, stats_part, stats_rows);                                   \
    } while (0)
    if (bf3) {
        if (a_scale) CMVAE_MULTI_BF3(true); else CMVAE_MULTI_BF3(false);
        return check_launch("conv_gemm_wstat_multi_kernel (bf16 pipe)");
    }
#undef CMVAE_MULTI_BF3
#define CMVAE_MULTI(HAS_T, PM)                                                                                         \
    do {                                                                                                               \
        static const bool opted = [] {                                                                                 \
            (void)hipFuncSetAttribute((const void*)conv_gemm_wstat_multi_kernel<0, HAS_T, PM>,                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                         \
            (void)hipGetLastError();                                                                                   \
            return true;                                                                                               \
        }();                                                                                                           \
        (void)opted;                                                                                                   \
        hipLaunchKernelGGL((conv_gemm_wstat_multi_kernel<0, HAS_T, PM>), dim3(used), dim3(256), lds, st, p, A, Wt,     \
                           bias, a_scale, a_shift, out, stats_part, stats_rows);                                       \
    } while (0)
    if (a_scale) { if (pm) CMVAE_MULTI(true, true); else CMVAE_MULTI(true, false); }
    else { if (pm) CMVAE_MULTI(false, true); else CMVAE_MULTI(false, false); }
#undef CMVAE_MULTI
    return check_launch("conv_gemm_wstat_multi_kernel");
}

extern "C" int cmvae_conv_gemm(const cmvae_conv_geom* geom, const float* A, const float* Wt, const float* bias,
                               const float* a_scale, const float* a_shift, int act, int n_col_blocks, float* out,
                               float* stats_part, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(geom && A && Wt && out, "cmvae_conv_gemm: NULL argument");
    CMVAE_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "cmvae_conv_gemm: scale and shift go together");
    CMVAE_REQUIRE(n_col_blocks >= 1 && act >= 0 && act <= 2, "cmvae_conv_gemm: bad act / column blocks");
    ConvGeom g = to_geom(geom);
    if (int e = check_geom(g)) return e;
    CMVAE_REQUIRE(g.o_pitch >= 64 * n_col_blocks, "cmvae_conv_gemm: output pitch < columns");
    CMVAE_REQUIRE(!stats_part || n_col_blocks == 1, "cmvae_conv_gemm: statistics need a single column block");
    if (g.rows_total == 0) return 0;
    dim3 grid(gemm_grid(g.rows_total, n_col_blocks), n_col_blocks);
    hipStream_t st = (hipStream_t)stream;
    const bool pm = g.pm_nimg > 0;
    CMVAE_REQUIRE(!pm || act == 0, "cmvae_conv_gemm: pixel-major launches support act = 0 only");
    {   // weight-stationary, barrier-free kernel wherever the launch's weight tiles fit LDS
        bool handled = false;
        int e = 0;
        const int srows = (int)grid.x;
#define CMVAE_WSTAT(ACT, PM)                                                                                          \
        e = a_scale ? dispatch_wstat<ACT, true, PM>(g, A, Wt, bias, a_scale, a_shift, out, stats_part, srows, n_col_blocks, st, &handled) \
                    : dispatch_wstat<ACT, false, PM>(g, A, Wt, bias, a_scale, a_shift, out, stats_part, srows, n_col_blocks, st, &handled)
        if (pm) { CMVAE_WSTAT(0, true); }
        else switch (act) {
            case 0: CMVAE_WSTAT(0, false); break;
            case 1: CMVAE_WSTAT(1, false); break;
            default: CMVAE_WSTAT(2, false); break;
        }
#undef CMVAE_WSTAT
        if (handled || e) return e;
    }
    // (10 .. 18 taps, pixel-major, no fold — the decoder's first 4 x 4 layer's data gradient, 16 taps over 153 600 rows —
    // as bf16-pipe passes like cmvae_conv_gemm_bnbwd's: 33 + 38 + 40 + 35 us against 132 us here; not kept)
    if (!a_scale) {          // no folded BatchNorm: operand fragments straight from global memory
        if (pm) hipLaunchKernelGGL((conv_gemm_direct_kernel<0, false, true>), grid, dim3(256), 0, st, g, A, Wt, bias, out, stats_part, (const float*)nullptr, 1, 0);
        else switch (act) {
            case 0: hipLaunchKernelGGL((conv_gemm_direct_kernel<0, false, false>), grid, dim3(256), 0, st, g, A, Wt, bias, out, stats_part, (const float*)nullptr, 1, 0); break;
            case 1: hipLaunchKernelGGL((conv_gemm_direct_kernel<1, false, false>), grid, dim3(256), 0, st, g, A, Wt, bias, out, stats_part, (const float*)nullptr, 1, 0); break;
            default: hipLaunchKernelGGL((conv_gemm_direct_kernel<2, false, false>), grid, dim3(256), 0, st, g, A, Wt, bias, out, stats_part, (const float*)nullptr, 1, 0); break;
        }
        return check_launch("conv_gemm_direct_kernel");
    }
    CMVAE_REQUIRE(!pm, "cmvae_conv_gemm: a pixel-major launch with a folded BatchNorm needs <= %d taps", WSTAT_TAPS8);
#define CMVAE_LAUNCH_GEMM(ACT)                                                                                  \
    hipLaunchKernelGGL((conv_gemm_kernel<ACT, true, false>), grid, dim3(256), 0, st, g, A, Wt, bias, a_scale,   \
                       a_shift, out, stats_part)
    switch (act) {
        case 0: CMVAE_LAUNCH_GEMM(0); break;
        case 1: CMVAE_LAUNCH_GEMM(1); break;
        default: CMVAE_LAUNCH_GEMM(2); break;
    }
#undef CMVAE_LAUNCH_GEMM
    return check_launch("conv_gemm_kernel");
}

extern "C" int cmvae_conv_gemm_bnbwd(const cmvae_conv_geom* geom, const float* A, const float* Wt, float* out,
                                     const float* x, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, float* sums_part, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(geom && A && Wt && out && x && scale && shift && mean && invstd && sums_part,
                  "cmvae_conv_gemm_bnbwd: NULL argument");
    ConvGeom g = to_geom(geom);
    if (int e = check_geom(g)) return e;
    CMVAE_REQUIRE(g.o_pitch == 64, "cmvae_conv_gemm_bnbwd: the output (and x) must be [pixels, 64]");
    const int gx = gemm_grid(g.rows_total > 0 ? g.rows_total : 1, 1);
    if (g.rows_total == 0) {
        (void)hipMemsetAsync(sums_part, 0, (size_t)gx * 128 * sizeof(float), (hipStream_t)stream);
        return check_launch("cmvae_conv_gemm_bnbwd (empty)");
    }
    hipStream_t st = (hipStream_t)stream;
    const BnBwd bnb{x, scale, shift, mean, invstd};
    const float* none = nullptr;
    // large pixel-major launches: products on the bf16 pipe (mfma_split6), at most BF3_TAPS taps' weight planes in LDS
    // per pass, later passes adding to the output; the last one carries the BatchNorm-backward sums
    if (g.pm_nimg > 0 && g.rows_total >= bf3_min_rows() && g.ntaps <= 3 * BF3_TAPS) {
        if (g.ntaps <= BF3_TAPS)
            return launch_wstat<0, false, 8, true, true, false, true>(g, A, Wt, none, none, none, out, sums_part, gx, 1, 0, g.ntaps, st, bnb);
        // More taps than fit LDS: passes over tap groups, the later ones adding to the output (a tile with no in-image
        // tap in a pass adds zero from one masked tap: tile_info), the last one with the BatchNorm-backward sums.
        // The passes are bound by HBM traffic (16 taps as 5 + 5 + 6 in tap order: 401 / 538 / 621 us for 2.0 / 2.65 /
        // 3.3 GB, 5 TB/s each), and with stride 2 a tap reads gradient pixels of ONE (row, column) parity class: a pass
        // whose taps mix classes reads most of the gradient tensor again.  Taps are therefore ordered by class and the
        // passes cut at class boundaries where that fits: a 4 x 4 kernel = four passes of four taps, each reading
        // its quarter of the gradient once (330 + 389 + 365 + 462 us; same-box step 11.14 -> 10.85 ms together with the
        // earlier request of the output in adding passes); layer 2's 3 x 3 kernel = (odd, odd) + (odd, even) taps, then
        // the other three: 139 + 182 us against 330 in one fp32 launch (in 4 + 5 taps of tap order, and with the
        // output requested in the epilogue, it had lost: 347 us).
        ConvGeom gp;
        int cut[8];                                               // pass p = taps cut[p] .. cut[p + 1] - 1 of gp
        const int ncut = bf3_pass_plan(g, gp, cut);
        for (int p = 0; p < ncut - 1; ++p) {
            const int t0 = cut[p], nt = cut[p + 1] - cut[p];
            if (p == 0) { if (int e = launch_wstat<0, false, 8, true, false, false, true>(gp, A, Wt, none, none, none, out, nullptr, 0, 1, t0, nt, st)) return e; }
            else if (int e = launch_wstat<0, false, 8, true, false, true, true>(gp, A, Wt, none, none, none, out, nullptr, 0, 1, t0, nt, st)) return e;
        }
        return launch_wstat<0, false, 8, true, true, true, true>(gp, A, Wt, none, none, none, out, sums_part, gx, 1, cut[ncut - 1], g.ntaps - cut[ncut - 1], st, bnb);
    }
    // (below the bf16 pipe's row count: 7 .. 9 taps fit LDS at once as fp32)
    if (g.pm_nimg > 0 && g.ntaps > BF3_TAPS && g.ntaps <= WSTAT_TAPS8 && g.rows_total >= 3 * 256 * 256)
        return launch_wstat<0, false, 8, true, true>(g, A, Wt, none, none, none, out, sums_part, gx, 1, 0, g.ntaps, st, bnb);
    const dim3 grid(gx, 1);
    if (g.pm_nimg > 0)
        hipLaunchKernelGGL((conv_gemm_direct_kernel<0, false, true, true>), grid, dim3(256), 0, st, g, A, Wt, none, out, sums_part, none, 1, 0, bnb);
    else
        hipLaunchKernelGGL((conv_gemm_direct_kernel<0, false, false, true>), grid, dim3(256), 0, st, g, A, Wt, none, out, sums_part, none, 1, 0, bnb);
    return check_launch("conv_gemm_direct_kernel (bn backward sums)");
}

extern "C" size_t cmvae_conv_wgrad_workspace_bytes(int rows_total) {
    (void)rows_total;
    const int nblk = wgrad_grid(1 << 30);      // sized for the full grid: column/tap blocks share it
    return align_up((size_t)nblk * 4 * 4096 * sizeof(float)) + align_up((size_t)nblk * 64 * sizeof(float));
}

// ny 64-column blocks of G (dW advances by y_stride per block) x the taps of g, in as few launches as possible
static int wgrad_launch(const ConvGeom& g, const float* A, const float* a_scale, const float* a_shift,
                        const float* G, int g_pitch, int g_col, const int64_t* tap_off_host, int64_t s_ci,
                        int64_t s_co, float* dW, float* db, int accumulate_db, void* workspace, hipStream_t st,
                        int ny = 1, int64_t y_stride = 0) {
    const int nt_launch = g.ntaps < 4 ? g.ntaps : 4;
    const int nz_all = g.ntaps % nt_launch == 0 ? g.ntaps / nt_launch : 1;       // tap groups in grid.z when uniform
    const int max_blocks = wgrad_grid(1 << 30);
    int nblk = max_blocks / (ny * nz_all);
    const int ntiles = (g.rows_total + BM - 1) / BM;
    if (nblk < 1) nblk = 1;
    if (nblk > ntiles) nblk = ntiles;
    float* part = (float*)workspace;
    float* db_part = (float*)((char*)workspace + align_up((size_t)max_blocks * 4 * 4096 * sizeof(float)));
    if ((long)nblk * ny * nz_all > max_blocks) return set_error("wgrad: %d x %d column/tap blocks exceed the workspace", ny, nz_all);
    {   // 5 .. 16 taps of one gradient column block (the 3x3 convs: 4 + 4 + 1): the tap groups side by side in one
        // launch and one reduction, workgroups in proportion to each group's tap work
        const int npix = g.Hg * g.Wg, nimg = g.rows_total / npix;
        if (ny == 1 && g_col == 0 && g.ntaps > 4 && g.ntaps <= 16 && g.ntaps % 4 != 0 &&
            g.rows_total % npix == 0 && nimg % 16 == 0 && (long)nimg * g.Ho * g.Wo * g_pitch < (1L << 31)) {
            ConvGeomMulti p{};
            WgradMultiOut mo{};
            mo.s_ci = s_ci; mo.s_co = s_co;
            const int ngrp = (g.ntaps + 3) / 4;
            const long chunks = (long)(nimg / 16) * npix;
            double wk[4] = {0., 0., 0., 0.}, wsum = 0.;
            for (int k = 0; k < ngrp; ++k) {
                const int t0 = 4 * k, nt = g.ntaps - t0 < 4 ? g.ntaps - t0 : 4;
                p.g[k] = g;
                p.g[k].ntaps = nt;
                for (int t = 0; t < nt; ++t) {
                    p.g[k].dy[t] = g.dy[t0 + t]; p.g[k].dx[t] = g.dx[t0 + t];
                    p.g[k].a_col[t] = g.a_col[t0 + t]; p.g[k].w_off[t] = g.w_off[t0 + t];
                    mo.tap_off[4 * k + t] = tap_off_host[t0 + t];
                }
                mo.ntaps[k] = nt;
                wk[k] = (double)chunks / (nt > 2 ? 2 : 4) * (nt == 1 ? 0.6 : 1.0);
                wsum += wk[k];
            }
            int used = 0;
            for (int k = 0; k <= 4; ++k) {
                p.begin[k] = mo.begin[k] = used;
                if (k >= ngrp) { if (k < 4) p.g[k] = p.g[0]; continue; }
                long n = (long)(max_blocks * wk[k] / wsum);
                if (n > chunks) n = chunks;
                if (n < 1) n = 1;
                used += (int)n;
            }
            float* dbp = db ? db_part : nullptr;
            const int db_blocks = p.begin[1];          // every group reads the same gradient rows: group 0 sums them
            if (a_scale) hipLaunchKernelGGL(conv_wgrad_stream_multi_kernel<true>, dim3(used), dim3(256), 0, st, p, nimg, db_blocks, A, a_scale, a_shift, G, g_pitch, part, dbp);
            else hipLaunchKernelGGL(conv_wgrad_stream_multi_kernel<false>, dim3(used), dim3(256), 0, st, p, nimg, db_blocks, A, a_scale, a_shift, G, g_pitch, part, dbp);
            if (int e = check_launch("conv_wgrad_stream_multi_kernel")) return e;
            hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(4 * 64 + 1, ngrp), dim3(1024), 0, st, part, mo, dW, dbp, db, db_blocks, accumulate_db);
            return check_launch("wgrad_reduce_multi_kernel");
        }
    }
    WgradOut o{};
    o.s_ci = s_ci; o.s_co = s_co; o.y_stride = y_stride;
    for (int t0 = 0; t0 < g.ntaps; t0 += nt_launch * nz_all) {
        const int nt = g.ntaps - t0 < nt_launch ? g.ntaps - t0 : nt_launch;
        const int nz = nz_all > 1 ? nz_all : 1;
        float* dbp = (db && t0 == 0) ? db_part : nullptr;
        const dim3 grid(nblk, ny, nz);
        const bool pm = g.pm_nimg > 0;
        const int npix = g.Hg * g.Wg, nimg = g.rows_total / npix;
        // (the streaming kernel addresses the gradient with 32-bit element offsets, like the operand)
        if (g.rows_total % npix == 0 && nimg % 16 == 0 &&
            (long)nimg * g.Ho * g.Wo * g_pitch < (1L << 31)) {
            if (a_scale) hipLaunchKernelGGL(conv_wgrad_stream_kernel<true>, grid, dim3(256), 0, st, g, t0, nt, nimg, A, a_scale, a_shift, G, g_pitch, g_col, part, dbp);
            else hipLaunchKernelGGL(conv_wgrad_stream_kernel<false>, grid, dim3(256), 0, st, g, t0, nt, nimg, A, a_scale, a_shift, G, g_pitch, g_col, part, dbp);
        } else {
#define CMVAE_WGRAD(NT, HAS_T, PM) \
        hipLaunchKernelGGL((conv_wgrad_kernel<NT, HAS_T, PM>), grid, dim3(256), 0, st, g, t0, A, a_scale, a_shift, G, g_pitch, g_col, part, dbp)
#define CMVAE_WGRAD_NT(NT)                                                     \
        if (a_scale) { if (pm) CMVAE_WGRAD(NT, true, true); else CMVAE_WGRAD(NT, true, false); } \
        else { if (pm) CMVAE_WGRAD(NT, false, true); else CMVAE_WGRAD(NT, false, false); }
        switch (nt) {
            case 1: CMVAE_WGRAD_NT(1) break;
            case 2: CMVAE_WGRAD_NT(2) break;
            case 3: CMVAE_WGRAD_NT(3) break;
            default: CMVAE_WGRAD_NT(4) break;
        }
#undef CMVAE_WGRAD_NT
#undef CMVAE_WGRAD
        }
        if (int e = check_launch("conv_wgrad_kernel")) return e;
        for (int t = 0; t < nt * nz; ++t) o.tap_off[t] = tap_off_host[t0 + t];
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nt * 64 + 1, ny, nz), dim3(1024), 0, st, part, nblk, nt, o, dW,
                           dbp, dbp ? db : nullptr, accumulate_db);
        if (int e = check_launch("wgrad_reduce_kernel")) return e;
    }
    return 0;
}

extern "C" int cmvae_conv_wgrad(const cmvae_conv_geom* geom, const float* A, const float* a_scale,
                                const float* a_shift, const float* G, int g_pitch, int g_col,
                                const int64_t* tap_off_host, int64_t s_ci, int64_t s_co, float* dW, float* db,
                                int accumulate_db, void* workspace, size_t workspace_bytes, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(geom && A && G && tap_off_host && dW && workspace, "cmvae_conv_wgrad: NULL argument");
    CMVAE_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "cmvae_conv_wgrad: scale and shift go together");
    ConvGeom g = to_geom(geom);
    if (int e = check_geom(g)) return e;
    CMVAE_REQUIRE(g_pitch >= 64 && g_col >= 0 && g_col + 64 <= g_pitch && g_pitch % 4 == 0 && g_col % 4 == 0,
                  "cmvae_conv_wgrad: bad gradient pitch/column");
    CMVAE_REQUIRE(workspace_bytes >= cmvae_conv_wgrad_workspace_bytes(g.rows_total), "cmvae_conv_wgrad: workspace too small");
    if (g.rows_total == 0) return 0;
    return wgrad_launch(g, A, a_scale, a_shift, G, g_pitch, g_col, tap_off_host, s_ci, s_co, dW, db, accumulate_db,
                        workspace, (hipStream_t)stream);
}

extern "C" int cmvae_conv_wgrad_multi(const cmvae_conv_geom* geoms, int ngeom, const float* A, const float* a_scale,
                                      const float* a_shift, const float* G, int g_pitch,
                                      const int64_t* tap_off_host, int64_t s_ci, int64_t s_co, float* dW, float* db,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(geoms && A && G && tap_off_host && dW && workspace, "cmvae_conv_wgrad_multi: NULL argument");
    CMVAE_REQUIRE(ngeom >= 1 && ngeom <= 4, "cmvae_conv_wgrad_multi: %d geometries (1..4)", ngeom);
    CMVAE_REQUIRE((a_scale == nullptr) == (a_shift == nullptr), "cmvae_conv_wgrad_multi: scale and shift go together");
    CMVAE_REQUIRE(g_pitch >= 64 && g_pitch % 4 == 0, "cmvae_conv_wgrad_multi: bad gradient pitch");
    CMVAE_REQUIRE(workspace_bytes >= cmvae_conv_wgrad_workspace_bytes(0), "cmvae_conv_wgrad_multi: workspace too small");
    ConvGeomMulti p{};
    hipStream_t st = (hipStream_t)stream;
    bool streamable = true;
    int nimg = -1;
    for (int k = 0; k < ngeom; ++k) {
        p.g[k] = to_geom(&geoms[k]);
        const ConvGeom& g = p.g[k];
        if (int e = check_geom(g)) return e;
        CMVAE_REQUIRE(g.ntaps <= 4, "cmvae_conv_wgrad_multi: geometry %d has %d taps (at most 4)", k, g.ntaps);
        const int npix = g.Hg * g.Wg, ni = g.rows_total / npix;
        if (nimg < 0) nimg = ni;
        CMVAE_REQUIRE(ni == nimg && g.Ho == p.g[0].Ho && g.Wo == p.g[0].Wo && g.Hi == p.g[0].Hi && g.Wi == p.g[0].Wi &&
                      g.a_pitch == p.g[0].a_pitch,
                      "cmvae_conv_wgrad_multi: geometry %d is not over the same tensors as geometry 0", k);
        streamable = streamable && ni > 0 && ni % 16 == 0 && (long)ni * g.Ho * g.Wo * g_pitch < (1L << 31);
    }
    if (nimg == 0) return 0;
    if (!streamable) {        // image counts the streaming kernel does not take: one launch per geometry
        const int64_t* to = tap_off_host;
        for (int k = 0; k < ngeom; ++k) {
            if (int e = wgrad_launch(p.g[k], A, a_scale, a_shift, G, g_pitch, 0, to, s_ci, s_co, dW, db, k > 0,
                                     workspace, st)) return e;
            to += p.g[k].ntaps;
        }
        return 0;
    }
    // workgroups per geometry in proportion to the time of one of its row streams: chunks / streams per workgroup,
    // a lone tap costing about 0.6 of a pair
    // large launches of four-tap phases (two operand rows x {unshifted, shifted by one column} in that order, same
    // operand columns) whose image count is a multiple of 32: products on the bf16 pipe, one workgroup per CU
    long all_rows = 0;
    bool rows4 = true;
    for (int k = 0; k < ngeom; ++k) {
        const ConvGeom& g = p.g[k];
        all_rows += g.rows_total;
        rows4 = rows4 && g.ntaps == 4 && g.sx == 1 && g.Wg == g.Wi;
        for (int t = 0; rows4 && t < 4; t += 2) {
            const int d0 = g.dx[t], d1 = g.dx[t + 1];
            rows4 = g.dy[t] == g.dy[t + 1] && g.a_col[t] == g.a_col[t + 1] &&
                    ((d0 == 0 && (d1 == 1 || d1 == -1)) || (d1 == 0 && (d0 == 1 || d0 == -1)));
        }
    }
    // (its requests are buffer loads: 32-bit byte offsets, so both tensors must stay below 4 GB)
    const bool bf3 = rows4 && nimg % 32 == 0 && all_rows >= (long)bf3_min_rows() &&
                     (long)nimg * p.g[0].Ho * p.g[0].Wo * g_pitch * 4 < (1L << 32) &&
                     (long)nimg * p.g[0].Hi * p.g[0].Wi * p.g[0].a_pitch * 4 < (1L << 32);
    const int max_blocks = bf3 ? compute_units() : wgrad_grid(1 << 30);
    double wsum = 0., wk[4] = {0., 0., 0., 0.};
    long chunks[4] = {0, 0, 0, 0};
    for (int k = 0; k < ngeom; ++k) {
        const ConvGeom& g = p.g[k];
        chunks[k] = (long)(nimg / (bf3 ? 32 : 16)) * g.Hg * g.Wg;
        wk[k] = (double)chunks[k] / (g.ntaps > 2 ? 2 : 4) * (g.ntaps == 1 ? 0.6 : 1.0);
        wsum += wk[k];
    }
    WgradMultiOut o{};
    o.s_ci = s_ci; o.s_co = s_co;
    int used = 0;
    const int64_t* to = tap_off_host;
    for (int k = 0; k <= 4; ++k) {
        p.begin[k] = o.begin[k] = used;         // (of an unused geometry: the grid size, never reached)
        if (k >= ngeom) { if (k < 4) p.g[k] = p.g[0]; continue; }
        long n = (long)(max_blocks * wk[k] / wsum);
        if (n > chunks[k]) n = chunks[k];
        if (n < 1) n = 1;
        used += (int)n;
        o.ntaps[k] = p.g[k].ntaps;
        for (int t = 0; t < p.g[k].ntaps; ++t) o.tap_off[4 * k + t] = *to++;
    }
    CMVAE_REQUIRE(used <= max_blocks, "cmvae_conv_wgrad_multi: %d workgroups exceed the workspace", used);
    float* part = (float*)workspace;
    float* db_part = (float*)((char*)workspace + align_up((size_t)wgrad_grid(1 << 30) * 4 * 4096 * sizeof(float)));
    float* dbp = db ? db_part : nullptr;
    if (bf3) {
        if (a_scale) hipLaunchKernelGGL(conv_wgrad_bf3_multi_kernel<true>, dim3(used), dim3(256), 0, st, p, nimg, used, A, a_scale, a_shift, G, g_pitch, part, dbp);
        else hipLaunchKernelGGL(conv_wgrad_bf3_multi_kernel<false>, dim3(used), dim3(256), 0, st, p, nimg, used, A, a_scale, a_shift, G, g_pitch, part, dbp);
        if (int e = check_launch("conv_wgrad_bf3_multi_kernel")) return e;
        hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(4 * 64 + 1, ngeom), dim3(1024), 0, st, part, o, dW, dbp, db, used, 0);
        return check_launch("wgrad_reduce_multi_kernel");
    }
    if (a_scale) hipLaunchKernelGGL(conv_wgrad_stream_multi_kernel<true>, dim3(used), dim3(256), 0, st, p, nimg, used, A, a_scale, a_shift, G, g_pitch, part, dbp);
    else hipLaunchKernelGGL(conv_wgrad_stream_multi_kernel<false>, dim3(used), dim3(256), 0, st, p, nimg, used, A, a_scale, a_shift, G, g_pitch, part, dbp);
    if (int e = check_launch("conv_wgrad_stream_multi_kernel")) return e;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(4 * 64 + 1, ngeom), dim3(1024), 0, st, part, o, dW, dbp, db, used, 0);
    return check_launch("wgrad_reduce_multi_kernel");
}

// ============================================================================ nn.Linear on the same kernels
// y = act(x W^T + b): one "image" per row, in_f/64 taps over the operand columns, out_f/64 column blocks.
namespace cmvae {

static ConvGeom linear_geom(int rows, int K, int N, int a_pitch, int o_pitch) {
    ConvGeom g{};
    g.rows_total = rows;
    g.Hg = g.Wg = g.Hi = g.Wi = g.sy = g.sx = g.Ho = g.Wo = g.oys = g.oxs = 1;
    g.ntaps = K / 64;
    g.a_pitch = a_pitch; g.o_pitch = o_pitch; g.w_pitch = K;
    g.w_nblk_stride = 64L * K;
    for (int t = 0; t < g.ntaps; ++t) { g.a_col[t] = 64 * t; g.w_off[t] = 64 * t; }
    return g;
}

// A Linear layer with many input blocks and few outputs (the data gradient of the causal layer's fc1: 2048 -> 64 over
// 25 600 rows) has one column block and 100 row tiles: on the direct kernel 200 workgroups of one wave per SIMD walk 32
// taps each between barriers (168 us, 40 TFLOP/s).  Split over the contraction instead: tap group y (<= 6 taps of
// bf16 weight planes in LDS) of the weight-stationary kernel writes its partial product to part[y], and the groups are
// added in fixed order (reproducible, no atomics): 104 + 12 us for the same launch, 3 tiles x 6 taps per workgroup at
// ~5 us per tap step -- the 8 KB operand rows are read 256 bytes at a time with 64 KB in flight per CU (starting every
// workgroup at another tap, so that all 32 column blocks are read at any moment, changed nothing: 104 vs 107 us).
__global__ __launch_bounds__(512, 1) void conv_gemm_wstat_splitk_kernel(ConvGeom g, const float* __restrict__ A,
                                                                        const float* __restrict__ Wt,
                                                                        float* __restrict__ part, long part_stride) {
    const int ngrp = gridDim.y, base = g.ntaps / ngrp, rem = g.ntaps - base * ngrp, y = blockIdx.y;
    const int tap0 = y * base + (y < rem ? y : rem), nt = base + (y < rem ? 1 : 0);
    wstat_body<0, false, 8, false, false, false, true>(g, A, Wt, nullptr, nullptr, nullptr, part + y * part_stride, nullptr,
                                                        0, tap0, nt, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.z,
                                                        (int)blockIdx.x, (int)gridDim.x);
}
// out[r][c] = sum_y part[y][r][c] (y ascending), c < ncols (a multiple of 4)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ngrp, long part_stride,
                                                            long n4, int ncols4, int o_pitch, float* __restrict__ out) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long)gridDim.x * 256) {
        float4 s = reinterpret_cast<const float4*>(part)[e];
        for (int y = 1; y < ngrp; ++y) {
            const float4 v = reinterpret_cast<const float4*>(part + y * part_stride)[e];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const long r = e / ncols4;
        *reinterpret_cast<float4*>(out + r * o_pitch + 4 * (e - r * ncols4)) = s;
    }
}
static int splitk_groups(int rows, int ntaps, int ncb) {
#if CMVAE_BF3_LINEAR
    if (ntaps <= BF3_TAPS || rows < CMVAE_BF3_LINEAR) return 0;
    const int ngrp = (ntaps + BF3_TAPS - 1) / BF3_TAPS;
    // worth it while the direct kernel's (row tile, column block) workgroups leave the chip half empty
    return (long)((rows + 127) / 128) * ncb <= 2L * compute_units() ? ngrp : 0;
#else
    return 0;
#endif
}
static size_t splitk_workspace_bytes(int rows, int ntaps, int ncb) {
    return align_up((size_t)splitk_groups(rows, ntaps, ncb) * rows * ncb * 64 * sizeof(float));
}
static int launch_gemm_splitk(const ConvGeom& g, const float* A, const float* Wt, int ncb, float* out, float* part,
                              hipStream_t st) {
    const int ngrp = splitk_groups(g.rows_total, g.ntaps, ncb);
    const size_t lds = (size_t)((g.ntaps + ngrp - 1) / ngrp) * BF3_TAP_BYTES + 2 * 8 * 32 * sizeof(int) + 8 * 128 * sizeof(float);
    static const bool opted = [] {
        (void)hipFuncSetAttribute((const void*)conv_gemm_wstat_splitk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        (void)hipGetLastError();
        return true;
    }();
    (void)opted;
    ConvGeom gp = g;
    gp.o_pitch = 64 * ncb;                                         // partials are dense [rows][64 ncb]
    const long part_stride = (long)g.rows_total * gp.o_pitch;
    const int ntiles = (g.rows_total + 255) / 256;
    int nx = compute_units() / (ngrp * ncb);
    if (nx < 1) nx = 1;
    if (nx > ntiles) nx = ntiles;
    hipLaunchKernelGGL(conv_gemm_wstat_splitk_kernel, dim3(nx, ngrp, ncb), dim3(512), lds, st, gp, A, Wt, part, part_stride);
    if (int e = check_launch("conv_gemm_wstat_splitk_kernel")) return e;
    const long n4 = part_stride / 4;
    const int nb = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, st, part, ngrp, part_stride, n4, gp.o_pitch / 4,
                       g.o_pitch, out);
    return check_launch("splitk_reduce_kernel");
}

static int launch_gemm(const ConvGeom& g, const float* A, const float* Wt, const float* bias, int act, int ncb,
                       float* out, hipStream_t st, const float* mse_target = nullptr, int mse_M = 1, int mse_pitch = 0,
                       float* mse_part = nullptr) {
#if CMVAE_BF3_LINEAR
    // in_features 64 .. 384 over many rows (proj, the (mu, logvar) head, causal fc1 and their data gradients at the
    // per-sample row counts): the column block's weight planes resident in LDS (weight-stationary kernel, 8 waves),
    // products on the bf16 pipe.  Same-box A/B: Omniglot 20-way step 11.33 -> 11.18 ms (64-input fc1 included: -0.04 more),
    // Mini-ImageNet step 1.451 -> 1.418 ms.
    if (!mse_target && g.ntaps <= BF3_TAPS && g.rows_total >= CMVAE_BF3_LINEAR && (g.ntaps >= 2 || ncb >= 8)) {
        const float* none = nullptr;
        switch (act) {
            case 0: return launch_wstat<0, false, 8, false, false, false, true>(g, A, Wt, bias, none, none, out, nullptr, 0, ncb, 0, g.ntaps, st);
            case 1: return launch_wstat<1, false, 8, false, false, false, true>(g, A, Wt, bias, none, none, out, nullptr, 0, ncb, 0, g.ntaps, st);
            default: return launch_wstat<2, false, 8, false, false, false, true>(g, A, Wt, bias, none, none, out, nullptr, 0, ncb, 0, g.ntaps, st);
        }
    }
#endif
    // (fp32 MFMA, below the bf16 pipe's row count:)
    // in_features = 64 into many column blocks over many rows (fc1 of the NOTEARS-MLP layer, 64 -> 2048): a step of the
    // direct kernel would be one 64-deep tap between two barriers; here a workgroup keeps its column block's weight
    // tile in LDS and walks the row tiles without a barrier (147 vs 176 us on [38400, 64]).  Measured slower or equal
    // for in_features = 256 (68 vs 66 us) and for the small launches (17-19 vs 15 us), which stay on the direct kernel.
    if (!mse_target && g.ntaps == 1 && ncb >= 8 && g.rows_total >= 8192) {
        const float* none = nullptr;
        switch (act) {
            case 0: return launch_wstat<0, false, 4, false>(g, A, Wt, bias, none, none, out, nullptr, 0, ncb, 0, g.ntaps, st);
            case 1: return launch_wstat<1, false, 4, false>(g, A, Wt, bias, none, none, out, nullptr, 0, ncb, 0, g.ntaps, st);
            default: return launch_wstat<2, false, 4, false>(g, A, Wt, bias, none, none, out, nullptr, 0, ncb, 0, g.ntaps, st);
        }
    }
    dim3 grid(gemm_grid(g.rows_total, ncb), ncb);
    switch (act) {
        case 0: hipLaunchKernelGGL((conv_gemm_direct_kernel<0, true, false>), grid, dim3(256), 0, st, g, A, Wt, bias, out, mse_part, mse_target, mse_M, mse_pitch); break;
        case 1: hipLaunchKernelGGL((conv_gemm_direct_kernel<1, true, false>), grid, dim3(256), 0, st, g, A, Wt, bias, out, mse_part, mse_target, mse_M, mse_pitch); break;
        default: hipLaunchKernelGGL((conv_gemm_direct_kernel<2, true, false>), grid, dim3(256), 0, st, g, A, Wt, bias, out, mse_part, mse_target, mse_M, mse_pitch); break;
    }
    return check_launch("conv_gemm_kernel (linear)");
}

// Wt2[i][o] = W[o][i]
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ W, int out_f, int in_f,
                                                        float* __restrict__ Wt2) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8) tile[r][tx] = W[(long)(o0 + r) * in_f + i0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) Wt2[(long)(i0 + r) * out_f + o0 + tx] = tile[tx][r];
}

// column sums of x [rows, cols] in two fixed-order stages: part[blk][cols], then out[cols]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int rows, int cols,
                                                             int pitch, float* __restrict__ part) {
    const int rpb = (rows + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
    for (int c = threadIdx.x; c < cols; c += 256) {
        float s0 = 0.f, s1 = 0.f;
        int r = r0;
        for (; r + 1 < r1; r += 2) {
            s0 += x[(long)r * pitch + c];
            s1 += x[(long)(r + 1) * pitch + c];
        }
        if (r < r1) s0 += x[(long)r * pitch + c];
        part[(long)blockIdx.x * cols + c] = s0 + s1;
    }
}
constexpr int COLSUM_BLOCKS = 256;

}  // namespace cmvae

static int check_linear(const char* who, int rows, int in_f, int out_f) {
    CMVAE_REQUIRE(rows >= 0 && in_f >= 64 && out_f >= 64 && in_f % 64 == 0 && out_f % 64 == 0 && in_f <= 2048 &&
                  out_f <= 2048, "%s: rows=%d in=%d out=%d (features must be multiples of 64, at most 2048)", who,
                  rows, in_f, out_f);
    return 0;
}

extern "C" int cmvae_linear_fwd(const float* X, int x_pitch, int rows, int in_f, int out_f, const float* W,
                                const float* bias, int act, float* Y, int y_pitch, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(X && W && Y, "cmvae_linear_fwd: NULL argument");
    if (int e = check_linear("cmvae_linear_fwd", rows, in_f, out_f)) return e;
    CMVAE_REQUIRE(x_pitch >= in_f && x_pitch % 4 == 0 && y_pitch >= out_f && act >= 0 && act <= 2,
                  "cmvae_linear_fwd: bad pitch / act");
    CMVAE_REQUIRE((long)rows * x_pitch < (1L << 31), "cmvae_linear_fwd: operand larger than 2^31 elements");
    if (rows == 0) return 0;
    return launch_gemm(linear_geom(rows, in_f, out_f, x_pitch, y_pitch), X, W, bias, act, out_f / 64, Y,
                       (hipStream_t)stream);
}

extern "C" int cmvae_linear_mse_partials(int rows, int out_f) {
    cmvae::clear_stale_error();
    if (out_f < 64) return 0;
    return gemm_grid(rows, out_f / 64) * (out_f / 64);
}

extern "C" int cmvae_linear_mse_fwd(const float* X, int x_pitch, int rows, int in_f, int out_f, const float* W,
                                    const float* bias, int act, const float* target, int M, float* Y, int y_pitch,
                                    float* loss_part, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(X && W && Y && target && loss_part, "cmvae_linear_mse_fwd: NULL argument");
    if (int e = check_linear("cmvae_linear_mse_fwd", rows, in_f, out_f)) return e;
    CMVAE_REQUIRE(x_pitch >= in_f && x_pitch % 4 == 0 && y_pitch >= out_f && act >= 0 && act <= 2,
                  "cmvae_linear_mse_fwd: bad pitch / act");
    CMVAE_REQUIRE(M >= 1 && rows >= 1 && rows % M == 0, "cmvae_linear_mse_fwd: rows=%d must be a positive multiple of M=%d",
                  rows, M);
    CMVAE_REQUIRE((long)rows * x_pitch < (1L << 31), "cmvae_linear_mse_fwd: operand larger than 2^31 elements");
    return launch_gemm(linear_geom(rows, in_f, out_f, x_pitch, y_pitch), X, W, bias, act, out_f / 64, Y,
                       (hipStream_t)stream, target, M, out_f, loss_part);
}

extern "C" size_t cmvae_linear_bwd_workspace_bytes(int rows, int in_f, int out_f) {
    return cmvae_conv_wgrad_workspace_bytes(rows) + align_up((size_t)in_f * out_f * sizeof(float)) +
           align_up((size_t)COLSUM_BLOCKS * out_f * sizeof(float)) + splitk_workspace_bytes(rows, out_f / 64, in_f / 64);
}

extern "C" int cmvae_linear_bwd(const float* X, int x_pitch, const float* W, const float* dZ, int dz_pitch, int rows,
                                int in_f, int out_f, float* dX, int dx_pitch, float* dW, float* db, void* workspace,
                                size_t workspace_bytes, void* stream) {
    cmvae::clear_stale_error();
    CMVAE_REQUIRE(X && W && dZ && dW && workspace, "cmvae_linear_bwd: NULL argument");
    if (int e = check_linear("cmvae_linear_bwd", rows, in_f, out_f)) return e;
    CMVAE_REQUIRE(x_pitch >= in_f && x_pitch % 4 == 0 && dz_pitch >= out_f && dz_pitch % 4 == 0 &&
                  (!dX || dx_pitch >= in_f), "cmvae_linear_bwd: bad pitch");
    CMVAE_REQUIRE((long)rows * x_pitch < (1L << 31) && (long)rows * dz_pitch < (1L << 31),
                  "cmvae_linear_bwd: operand larger than 2^31 elements");
    CMVAE_REQUIRE(workspace_bytes >= cmvae_linear_bwd_workspace_bytes(rows, in_f, out_f),
                  "cmvae_linear_bwd: workspace too small");
    CMVAE_REQUIRE(rows >= 1, "cmvae_linear_bwd: no rows");
    hipStream_t st = (hipStream_t)stream;
    char* wsb = (char*)workspace;
    void* wg_ws = wsb;
    float* Wt2 = (float*)(wsb + cmvae_conv_wgrad_workspace_bytes(rows));
    float* cs_part = (float*)((char*)Wt2 + align_up((size_t)in_f * out_f * sizeof(float)));
    const int kin = in_f / 64, kout = out_f / 64;
    if (dX) {       // dX = dZ @ W: contraction over the out_f columns of dZ, weights transposed to [in][out]
        hipLaunchKernelGGL(transpose_kernel, dim3(in_f / 32, out_f / 32), dim3(256), 0, st, W, out_f, in_f, Wt2);
        if (int e = check_launch("transpose_kernel")) return e;
        if (splitk_groups(rows, kout, kin)) {
            float* sk_part = (float*)((char*)cs_part + align_up((size_t)COLSUM_BLOCKS * out_f * sizeof(float)));
            if (int e = launch_gemm_splitk(linear_geom(rows, out_f, in_f, dz_pitch, dx_pitch), dZ, Wt2, kin, dX, sk_part, st))
                return e;
        } else if (int e = launch_gemm(linear_geom(rows, out_f, in_f, dz_pitch, dx_pitch), dZ, Wt2, nullptr, 0, kin, dX, st))
            return e;
    }
    // dW[o][i] = sum_r dZ[r][o] X[r][i]: the operand with more 64-column blocks supplies the taps (4 per launch)
    int64_t tap_off[32];
    const bool taps_over_in = kout * ((kin + 3) / 4) <= kin * ((kout + 3) / 4);
    if (taps_over_in) {
        const ConvGeom g = linear_geom(rows, in_f, 64, x_pitch, 64);
        for (int t = 0; t < kin; ++t) tap_off[t] = 64 * t;
        if (int e = wgrad_launch(g, X, nullptr, nullptr, dZ, dz_pitch, 0, tap_off, 1, in_f, dW, db, 0, wg_ws, st, kout,
                                 64L * in_f))
            return e;
    } else {
        const ConvGeom g = linear_geom(rows, out_f, 64, dz_pitch, 64);
        for (int t = 0; t < kout; ++t) tap_off[t] = 64L * t * in_f;
        if (int e = wgrad_launch(g, dZ, nullptr, nullptr, X, x_pitch, 0, tap_off, in_f, 1, dW, nullptr, 0, wg_ws, st,
                                 kin, 64))
            return e;
        if (db) {
            const int nb = rows < COLSUM_BLOCKS ? rows : COLSUM_BLOCKS;
            hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb), dim3(256), 0, st, dZ, rows, out_f, dz_pitch, cs_part);
            if (int e = check_launch("colsum_partial_kernel")) return e;
            if (int e = cmvae_colsum_reduce(cs_part, nb, out_f, db, stream)) return e;
        }
    }
    return 0;
}

#if CMVAE_ABLATE == 30 || CMVAE_ABLATE == 73 || CMVAE_ABLATE == 74
extern "C" int cmvae_debug_wg_times(unsigned long long* out_host, int n_wg) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(cmvae::g_wg_times), (size_t)n_wg * 24) == hipSuccess ? 0 : -1;
}
extern "C" int cmvae_debug_wg_quarters(unsigned long long* out_host, int n_wg) {
    return hipMemcpyFromSymbol(out_host, HIP_SYMBOL(cmvae::g_wg_quarters), (size_t)n_wg * 32) == hipSuccess ? 0 : -1;
}
#endif
#if CMVAE_ABLATE == 20
extern "C" int cmvae_debug_clock_probe(long long* out2) {
    return hipMemcpyFromSymbol(out2, HIP_SYMBOL(cmvae::g_clock_probe), 16) == hipSuccess ? 0 : -1;
}
#endif
