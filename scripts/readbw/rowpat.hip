// Access-pattern probe for the direct (no LDS stage) shared sweep of 16-bit rows: a wave walks tiles of 16 rows of
// ROWB bytes (linear layout, pitch = ROWB) with 16-byte loads per lane, one block of `waves` waves per CU, tiles
// interleaved across all waves, a ring of loads in flight -- loads only, no arithmetic.  The lane -> (row, chunk)
// mapping is what varies:
//   0  row = L & 15, chunk = L >> 4: 64 B of each of 16 rows per instruction (the MFMA B-operand layout, as shipped)
//   1  chunk = L & 3, row = L >> 2:  the same 64-byte segments, the lanes of a row adjacent
//   2  row = L & 7, chunk = L >> 3:  128 B of each of 8 rows per instruction (two instructions per 16 rows)
//   3  chunk = L & 7, row = L >> 3:  the same 128-byte segments, the lanes of a row adjacent
// Build: hipcc --offload-arch=gfx950 -O3 -o rowpat rowpat.hip ;  ./rowpat [rows] [rowb] [waves] [ring]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, int D>
__global__ void rowpat_kernel(const uint8_t *p, size_t n_rows, int rowb, uint32_t *out)
{
    constexpr int SEG = MODE < 2 ? 64 : 128;
    constexpr int RPI = 1024 / SEG;  // rows per instruction
    const int lane = threadIdx.x & 63;
    const size_t nw = blockDim.x >> 6;
    const size_t wave = (size_t)blockIdx.x * nw + (threadIdx.x >> 6);
    const size_t stride = (size_t)gridDim.x * nw;
    const size_t n_units = n_rows / RPI;  // units of RPI rows
    const int S = rowb / SEG;             // instructions per unit
    int row, chunk;
    if (MODE == 0) { row = lane & 15; chunk = lane >> 4; }
    else if (MODE == 1) { chunk = lane & 3; row = lane >> 2; }
    else if (MODE == 2) { row = lane & 7; chunk = lane >> 3; }
    else { chunk = lane & 7; row = lane >> 3; }
    const size_t n_it = wave < n_units ? (n_units - wave + stride - 1) / stride : 0;
    const size_t NP = n_it * S;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 ring[D];
    size_t iu = wave;
    int is = 0;
    const uint8_t *ip = p + (wave * RPI + row) * (size_t)rowb + chunk * 16;
    auto issue = [&](int u) {
        ring[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(ip));
        if (++is == S) {
            is = 0;
            iu += stride;
            ip = p + ((iu < n_units ? iu : wave) * RPI + row) * (size_t)rowb + chunk * 16;
        } else {
            ip += SEG;
        }
    };
    size_t consumed = 0;
#pragma unroll
    for (int u = 0; u < D; u++) {
        issue(u);
        __builtin_amdgcn_sched_barrier(0);
    }
    while (consumed + 2 * D <= NP) {
#pragma unroll
        for (int u = 0; u < D; u++) {
            acc ^= ring[u];
            issue(u);
            __builtin_amdgcn_sched_barrier(0);
        }
        consumed += D;
    }
#pragma unroll
    for (int u = 0; u < D; u++) acc ^= ring[u];
    const uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (r == 0x12345678u) out[0] = r;
}

template <int MODE, int D>
int run(const uint8_t *buf, size_t n_rows, int rowb, int waves, uint32_t *out)
{
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount;
    auto launch = [&]() { hipLaunchKernelGGL((rowpat_kernel<MODE, D>), dim3(grid), dim3(64 * waves), 0, 0, buf, n_rows, rowb, out); };
    for (int i = 0; i < 3; i++) launch();
    CHK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; i++) launch();
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d ring %d waves %2d rows of %4d B: %.1f us/pass  %.2f TB/s\n", MODE, D, waves, rowb, ms / reps * 1e3,
           (double)n_rows * rowb / (ms / reps * 1e-3) / 1e12);
    return 0;
}

int main(int argc, char **argv)
{
    const size_t n_rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000;
    const int rowb = argc > 2 ? atoi(argv[2]) : 1536;
    uint8_t *buf;
    uint32_t *out;
    CHK(hipMalloc((void **)&buf, n_rows * rowb + 4096));
    CHK(hipMalloc((void **)&out, 64));
    CHK(hipMemset(buf, 0x5a, n_rows * rowb + 4096));
    for (int waves : {8, 12, 16}) {
        if (run<0, 3>(buf, n_rows, rowb, waves, out)) return 1;
        if (run<0, 6>(buf, n_rows, rowb, waves, out)) return 1;
        if (run<1, 3>(buf, n_rows, rowb, waves, out)) return 1;
        if (run<1, 6>(buf, n_rows, rowb, waves, out)) return 1;
        if (run<2, 4>(buf, n_rows, rowb, waves, out)) return 1;
        if (run<2, 6>(buf, n_rows, rowb, waves, out)) return 1;
        if (run<3, 4>(buf, n_rows, rowb, waves, out)) return 1;
        if (run<3, 6>(buf, n_rows, rowb, waves, out)) return 1;
    }
    return 0;
}
