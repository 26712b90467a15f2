// Ceiling probe: how fast can gfx950 stream-read a 3 GB buffer with 16-byte loads?
// Build: hipcc --offload-arch=gfx950 -O3 -o readbw readbw.hip
#include <hip/hip_runtime.h>
#include <unistd.h>
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


// tile walk of the shared sweeps: ONE block of WAVES waves per CU (grid = CUs), `lds` bytes of dynamic LDS
// allocated per block (the query image of the real kernel; touched once so that it is really allocated), every
// wave streams tiles of TILEB bytes KiB by KiB with a rotating ring of D loads, tiles interleaved across all waves.
typedef int v4i32 __attribute__((ext_vector_type(4)));

// WORK: 0 = loads only; 1 = + stage the LDS image from global memory at kernel start (as the sweeps do);
// 2 = + the int8 sweep's arithmetic per KiB step (6 A operands from LDS, 6 v_mfma_i32_16x16x64_i8, 8 v_dot4)
template <int D, int TILEB, int WORK>
__global__ void read_tile_kernel(const u32x4 *p, size_t n_tiles, uint32_t *out, int lds_words, const uint4 *image, int passes)
{
    extern __shared__ uint32_t dyn[];
    if ((WORK & 3) >= 1) {
        uint4 *dst = reinterpret_cast<uint4 *>(dyn);
        const int n16 = lds_words / 4;
        constexpr int U = 6;
        int i = threadIdx.x;
        for (; i + (U - 1) * (int)blockDim.x < n16; i += U * blockDim.x) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = image[i + u * blockDim.x];
#pragma unroll
            for (int u = 0; u < U; u++) dst[i + u * blockDim.x] = v[u];
        }
        for (; i < n16; i += blockDim.x) dst[i] = image[i];
    } else if (lds_words) {
        dyn[threadIdx.x % lds_words] = threadIdx.x;
    }
    constexpr int S = TILEB / 1024;  // KiB steps per tile
    const size_t lane0 = threadIdx.x & 63;
    const size_t lane = (WORK & 4) ? (lane0 & 15) * 4 + (lane0 >> 4) : lane0;  // 4: MFMA operand layout (16 lanes stride 64 B)
    const size_t nw = blockDim.x >> 6;
    const size_t wave = (size_t)blockIdx.x * nw + (threadIdx.x >> 6);
    const size_t stride = (size_t)gridDim.x * nw;
    const size_t n_it = wave < n_tiles ? (n_tiles - wave + stride - 1) / stride : 0;
    const size_t NP = n_it * S;
    u32x4 acc = {0, 0, 0, 0};
    v4i32 macc[6];
#pragma unroll
    for (int b = 0; b < 6; b++) macc[b] = v4i32{0, 0, 0, 0};
    int SQ = 0, SV = 0;
    const v4i32 *qimg = reinterpret_cast<const v4i32 *>(dyn);
    for (int pass = 0; pass < passes; pass++) {
        u32x4 ring[D];
        size_t itile = wave;
        int is = 0, cs = 0;
        const u32x4 *ip = p + wave * (TILEB / 16) + lane;
        auto issue = [&](int u) {
            ring[u] = __builtin_nontemporal_load(ip);
            if (++is == S) {
                is = 0;
                itile += stride;
                ip = p + (itile < n_tiles ? itile : wave) * (TILEB / 16) + lane;
            } else {
                ip += 64;
            }
        };
        auto consume = [&](int u) {
            if ((WORK & 3) < 2) {
                acc ^= ring[u];
                return;
            }
            const u32x4 v = ring[u];
            const uint32_t raw[4] = {v.x, v.y, v.z, v.w};
            v4i32 bop;
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int wn = (int)(raw[d] ^ 0x80808080u);
                bop[d] = wn;
                SQ = __builtin_amdgcn_sdot4(wn, wn, SQ, false);
                SV = __builtin_amdgcn_sdot4(wn, 0x01010101, SV, false);
            }
#pragma unroll
            for (int b = 0; b < 6; b++) {
                const v4i32 qc = qimg[(cs * 6 + b) * 64 + lane0];
                macc[b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(qc, bop, macc[b], 0, 0, 0);
            }
            if (++cs == S) cs = 0;
        };
        size_t consumed = 0;
#pragma unroll
        for (int u = 0; u < D; u++) {
            issue(u);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (pass == 0) __syncthreads();
        while (consumed + 2 * D <= NP) {
#pragma unroll
            for (int u = 0; u < D; u++) {
                consume(u);
                issue(u);
                __builtin_amdgcn_sched_barrier(0);
            }
            consumed += D;
        }
#pragma unroll
        for (int u = 0; u < D; u++) acc ^= ring[u];
    }
    uint32_t r = acc.x ^ acc.y ^ acc.z ^ acc.w;
#pragma unroll
    for (int b = 0; b < 6; b++) r ^= (uint32_t)(macc[b].x ^ macc[b].y ^ macc[b].z ^ macc[b].w);
    r ^= (uint32_t)(SQ ^ SV);
    if (r == 0x12345678u) out[0] = r + (lds_words ? dyn[0] : 0);
}

template <int D, int TILEB, int WORK>
int run_tile(const u32x4 *buf, size_t bytes, uint32_t *out, int waves, int blocks_per_cu, int lds_bytes, int cus, int passes = 1,
             bool per_launch_events = false, int gap_us = 0, bool own_stream = false)
{
    hipStream_t st = 0;
    if (own_stream) CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const size_t n_tiles = bytes / TILEB;
    auto kern = &read_tile_kernel<D, TILEB, WORK>;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    auto launch = [&]() {
        hipLaunchKernelGGL(kern, dim3(cus * blocks_per_cu), dim3(64 * waves), lds_bytes, st, buf, n_tiles, out, lds_bytes / 4,
                           reinterpret_cast<const uint4 *>(buf), passes);
    };
    for (int i = 0; i < 3; i++) launch();
    const int reps = 20;
    float ms = 0;
    if (per_launch_events) {  // as the library times its sweeps: events around every single launch
        for (int i = 0; i < reps; i++) {
            if (gap_us) usleep(gap_us);  // an idle card between launches, as between a caller's batches
            CHK(hipEventRecord(e0, st));
            launch();
            CHK(hipEventRecord(e1, st));
            CHK(hipEventSynchronize(e1));
            float m;
            CHK(hipEventElapsedTime(&m, e0, e1));
            ms += m;
        }
    } else {
        CHK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) launch();
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        CHK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("tiles %5d B ring %d %d blk/CU x %2d waves LDS %6d B work %d passes %d %s gap %d us%s: %.1f us/pass  %.2f TB/s\n", TILEB, D,
           blocks_per_cu, waves, lds_bytes, WORK, passes, per_launch_events ? "per-launch events" : "back-to-back     ", gap_us,
           own_stream ? " own stream" : "",
           ms / reps / passes * 1e3, (double)n_tiles * TILEB * passes / (ms / reps * 1e-3) / 1e12);
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
    if (argc > 3) {  // readbw <bytes> tiles random: pseudo-random bytes instead of a constant fill
        std::vector<uint32_t> h(1 << 22);
        uint64_t x = 88172645463325252ull;
        for (auto &w : h) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            w = (uint32_t)x;
        }
        for (size_t off = 0; off + h.size() * 4 <= n_vec * 16; off += h.size() * 4)
            CHK(hipMemcpy(reinterpret_cast<uint8_t *>(buf) + off, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("random fill\n");
    }
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, %zu bytes\n", prop.name, cus, bytes);
    if (argc > 2) {  // tile-walk matrix only: readbw <bytes> tiles
        const int img1 = 74 * 1024, img2 = 148 * 1024;  // one / two query groups of the 768-dim int8 image
        if (run_tile<4, 12288, 0>(buf, bytes, out, 12, 1, img2, cus, 2, true)) return 1;
        if (run_tile<4, 12288, 4>(buf, bytes, out, 12, 1, img2, cus, 2, true)) return 1;
        if (run_tile<4, 12288, 2>(buf, bytes, out, 12, 1, img2, cus, 2, true)) return 1;
        if (run_tile<4, 12288, 6>(buf, bytes, out, 12, 1, img2, cus, 2, true)) return 1;
        for (int gap : {0, 200, 1000, 5000})
            for (bool own : {false, true})
                if (run_tile<4, 12288, 6>(buf, bytes, out, 12, 1, img2, cus, 2, true, gap, own)) return 1;
        if (run_tile<4, 12288, 0>(buf, bytes, out, 12, 1, img2, cus, 2, true, 1000, true)) return 1;
        if (run_tile<4, 12288, 6>(buf, bytes, out, 8, 1, img2, cus, 2, true)) return 1;
        if (run_tile<4, 6144, 6>(buf, bytes, out, 12, 1, img2, cus, 2, true)) return 1;
        return 0;
    }
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
