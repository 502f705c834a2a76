// HBM throughput of the weight gradient's access pattern: every wave instruction reads 4 rows of RUN bytes (16 lanes x
// 16 bytes x RUN / 256 consecutive instructions) that lie IMG_STRIDE bytes apart — "32 images at one grid position" of a
// channels-last tensor — from a buffer far larger than the caches, every byte exactly once.  One wave per SIMD with
// DEPTH x 8 loads in flight (the kernel's register budget), or eight waves per CU.
//   RUN = 256: one pixel (what csrc/conv_gemm.hip's weight gradients read); 512 / 1024: 2 / 4 adjacent pixels per image.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// buffer: nimg images of img_bytes each; a "chunk" = 32 images x RUN bytes at byte offset pos * RUN of each image
template <int RUN, int DEPTH>
__global__ __launch_bounds__(256) void k(const char* __restrict__ buf, long img_bytes, int nblk_img, int npos,
                                         float* __restrict__ out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const long nchunks = (long)nblk_img * npos;
    const long nstreams = (long)gridDim.x * 4, sid = (long)blockIdx.x * 4 + w;
    const long c0 = nchunks * sid / nstreams, c1 = nchunks * (sid + 1) / nstreams;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int SUB = RUN / 256;               // 256-byte pieces per run: consecutive instructions walk them
    float4 r[DEPTH][8];
    auto issue = [&](int slot, long c) {
        const long ib = c / npos, pos = c - ib * npos;
        const char* p = buf + (ib * 32 + 8 * kq) * img_bytes + pos * 256 + 16 * i;      // pos counts 256-byte pixels
#pragma unroll
        for (int j = 0; j < 8; ++j) r[slot][j] = *reinterpret_cast<const float4*>(p + j * img_bytes);
    };
    // with RUN > 256 a wave takes SUB consecutive pixels of the same images one after the other (same DRAM pages)
    (void)SUB;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) if (c0 + d < c1) issue(d, c0 + d);
    for (long c = c0; c < c1; c += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (c + d >= c1) break;
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc.x += r[d][j].x; acc.y += r[d][j].y; acc.z += r[d][j].z; acc.w += r[d][j].w; }
            if (c + d + DEPTH < c1) issue(d, c + d + DEPTH);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int RUN, int DEPTH>
static void run(const char* buf, long img_bytes, int nimg, float* out, int blocks, const char* what) {
    // positions per image: img_bytes / 256; chunk order: positions fastest (a wave walks the pixels of its 32 images)
    const int npos = (int)(img_bytes / 256);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<RUN, DEPTH>), dim3(blocks), dim3(256), 0, 0, buf, img_bytes, nimg / 32, npos, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double bytes = (double)nimg * img_bytes;
    printf("%-44s image stride %7ld B, %4d workgroups, %d chunk(s) in flight: %7.3f ms  %6.2f TB/s\n", what, img_bytes, blocks,
           DEPTH, ms, bytes / ms * 1e-9);
}

int main() {
    const int nimg = 38400;
    const long img_x = 49 * 256, img_g = 196 * 256;            // layer 3: operand 7 x 7, gradient 14 x 14 pixels of 64 floats
    char* buf; float* out;
    CHECK(hipMalloc(&buf, (size_t)nimg * img_g));
    CHECK(hipMemset(buf, 0, (size_t)nimg * img_g));
    CHECK(hipMalloc(&out, 2048 * 256 * 4));
    run<256, 1>(buf, img_x, nimg, out, 256, "operand rows, 1 wave/SIMD");
    run<256, 2>(buf, img_x, nimg, out, 256, "operand rows, 1 wave/SIMD");
    run<256, 4>(buf, img_x, nimg, out, 256, "operand rows, 1 wave/SIMD");
    run<256, 1>(buf, img_x, nimg, out, 512, "operand rows, 2 waves/SIMD");
    run<256, 1>(buf, img_g, nimg, out, 256, "gradient rows, 1 wave/SIMD");
    run<256, 2>(buf, img_g, nimg, out, 256, "gradient rows, 1 wave/SIMD");
    run<256, 4>(buf, img_g, nimg, out, 256, "gradient rows, 1 wave/SIMD");
    run<256, 1>(buf, img_g, nimg, out, 512, "gradient rows, 2 waves/SIMD");
    run<256, 2>(buf, img_g, nimg, out, 1024, "gradient rows, 4 waves/SIMD");
    return 0;
}
