// Ceiling probe: how fast can gfx950 stream-read a 3 GB buffer with 16-byte loads?
// Build: hipcc --offload-arch=gfx950 -O3 -o readbw readbw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

template <bool NT, int D>
__global__ __launch_bounds__(256) void read_kernel(const u32x4 *p, size_t n_vec, uint32_t *out)
{
    // wave-interleaved: consecutive waves read consecutive 1 KB chunks, D loads in flight
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    u32x4 acc = {0, 0, 0, 0};
    size_t i = tid;
    for (; i + (D - 1) * stride < n_vec; i += D * stride) {
        u32x4 v[D];
#pragma unroll
        for (int u = 0; u < D; u++)
            v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < D; u++) acc ^= v[u];
    }
    for (; i < n_vec; i += stride) acc ^= p[i];
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[0] = r;  // never true for the fill below; keeps the loads alive
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// chunked: each wave owns D consecutive 1 KB chunks per step (a contiguous D KB run)
template <bool NT, int D>
__global__ __launch_bounds__(256) void read_chunk_kernel(const u32x4 *p, size_t n_vec, uint32_t *out)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t lane = threadIdx.x & 63;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t base = wave * (64 * D); base + 64 * D <= n_vec; base += n_waves * 64 * D) {
        u32x4 v[D];
#pragma unroll
        for (int u = 0; u < D; u++) v[u] = NT ? __builtin_nontemporal_load(p + base + u * 64 + lane) : p[base + u * 64 + lane];
#pragma unroll
        for (int u = 0; u < D; u++) acc ^= v[u];
    }
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[0] = r;
}

// segment pattern: a wave instruction reads SEG bytes of each of 64*16/SEG consecutive rows
// (lanes of a row group contiguous); a tile of rows is finished in ROWB/SEG instructions.
// SEG = 64, ROWB = 192: the single-query walk of 192-byte rows (config #5);
// SEG = 64, ROWB = 768: the MFMA operand layout of the shared sweep on 768-byte rows.
template <bool NT, int SEG, int ROWB>
__global__ __launch_bounds__(256) void read_seg_kernel(const uint8_t *p, size_t n_rows, uint32_t *out)
{
    constexpr int LPR = SEG / 16;        // lanes per row
    constexpr int RPW = 64 / LPR;        // rows per wave instruction
    constexpr int P = ROWB / SEG;        // instructions per tile
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t lane = threadIdx.x & 63;
    const size_t n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t g = lane / LPR, l = lane % LPR;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t tile = wave; (tile + 1) * RPW <= n_rows; tile += n_waves) {
        const uint8_t *rp = p + (tile * RPW + g) * ROWB + l * 16;
        u32x4 v[P];
#pragma unroll
        for (int i = 0; i < P; i++) {
            const u32x4 *q = reinterpret_cast<const u32x4 *>(rp + (size_t)i * SEG);
            v[i] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int i = 0; i < P; i++) acc ^= v[i];
    }
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[0] = r;
}

template <bool NT, int SEG, int ROWB>
int run_seg(const u32x4 *buf, size_t bytes, uint32_t *out, int bpc, int cus)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int grid = bpc * cus;
    const size_t n_rows = bytes / ROWB;
    auto launch = [&]() {
        hipLaunchKernelGGL((read_seg_kernel<NT, SEG, ROWB>), dim3(grid), dim3(256), 0, 0,
                           reinterpret_cast<const uint8_t *>(buf), n_rows, out);
    };
    for (int i = 0; i < 3; i++) launch();
    CHK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; i++) launch();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-6s segments of %3d B, rows of %4d B, blocks/CU=%d  %.1f us/pass  %.2f TB/s\n", NT ? "nt" : "plain",
           SEG, ROWB, bpc, ms / reps * 1e3, (double)n_rows * ROWB / (ms / reps * 1e-3) / 1e12);
    return 0;
}

template <bool NT, int D, bool CHUNK>
int run(const u32x4 *buf, size_t n_vec, uint32_t *out, int bpc, int cus, const char *name)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int grid = bpc * cus;
    auto launch = [&]() {
        if (CHUNK) hipLaunchKernelGGL((read_chunk_kernel<NT, D>), dim3(grid), dim3(256), 0, 0, buf, n_vec, out);
        else hipLaunchKernelGGL((read_kernel<NT, D>), dim3(grid), dim3(256), 0, 0, buf, n_vec, out);
    };
    for (int i = 0; i < 3; i++) launch();
    CHK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; i++) launch();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-10s %s D=%d blocks/CU=%d  %.1f us/pass  %.2f TB/s\n", name, CHUNK ? "chunk  " : "strided", D, bpc, ms / reps * 1e3,
           (double)n_vec * 16 / (ms / reps * 1e-3) / 1e12);
    return 0;
}

int main(int argc, char **argv)
{
    const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : 3072000000ull;
    const size_t n_vec = bytes / 16;
    u32x4 *buf;
    uint32_t *out;
    CHK(hipMalloc((void **)&buf, n_vec * 16));
    CHK(hipMalloc((void **)&out, 4));
    CHK(hipMemset(buf, 0x5a, n_vec * 16));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, %zu bytes\n", prop.name, cus, bytes);
    for (int bpc : {1, 2, 3, 4}) {
        if (run<true, 2, false>(buf, n_vec, out, bpc, cus, "nt")) return 1;
        if (run<true, 3, false>(buf, n_vec, out, bpc, cus, "nt")) return 1;
        if (run<true, 4, false>(buf, n_vec, out, bpc, cus, "nt")) return 1;
        if (run<true, 6, false>(buf, n_vec, out, bpc, cus, "nt")) return 1;
        if (run<false, 4, false>(buf, n_vec, out, bpc, cus, "plain")) return 1;
        if (run<true, 3, true>(buf, n_vec, out, bpc, cus, "nt")) return 1;
        if (run<true, 4, true>(buf, n_vec, out, bpc, cus, "nt")) return 1;
        if (run<true, 8, true>(buf, n_vec, out, bpc, cus, "nt")) return 1;
    }
    for (int bpc : {2, 3, 4}) {
        if (run_seg<false, 64, 192>(buf, bytes, out, bpc, cus)) return 1;
        if (run_seg<true, 64, 192>(buf, bytes, out, bpc, cus)) return 1;
        if (run_seg<false, 64, 768>(buf, bytes, out, bpc, cus)) return 1;
        if (run_seg<true, 64, 768>(buf, bytes, out, bpc, cus)) return 1;
        if (run_seg<true, 128, 768>(buf, bytes, out, bpc, cus)) return 1;
        if (run_seg<true, 256, 768>(buf, bytes, out, bpc, cus)) return 1;
    }
    CHK(hipFree(buf));
    CHK(hipFree(out));
    return 0;
}
