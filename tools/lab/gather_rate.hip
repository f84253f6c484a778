// Lab (not product): how many DIVERGENT 16-byte-per-lane loads can a CU's texture addresser / vector L1 take per second?
// k_paths issues 1.9e9 vector-memory wave-instructions per C2 frame (profiles/r05_full_bsdf_pmc.json: SQ_INSTS_VMEM), nearly
// all of them global_load_dwordx4 whose 64 lanes read 64 different 128-byte records (BVH nodes, triangle records).  This
// program times exactly that access shape at k_paths' occupancy (1 024 workgroups x 256 threads, 16 waves per CU):
//   K consecutive dwordx4 loads per record (K = 8: a 4-wide node, 3: a triangle record), records picked at random per lane
//   from a table of T bytes, `lanes` of 64 lanes active, `dep` = the next record index depends on the loaded data.
// Output: wave-instructions per microsecond per CU and the equivalent cycles per instruction at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O2 -o gather_rate tools/lab/gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int K, bool DEP>
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ table, unsigned n_records, int iters, int lanes,
                                                float *out) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    const bool on = (int)(threadIdx.x & 63) < lanes;
    unsigned idx = mix(tid) % n_records;
    float acc = 0.f;
    if (on) {
        for (int it = 0; it < iters; it++) {
            const float4 *q = table + (size_t)idx * 8;
            float4 v[K];
#pragma unroll
            for (int k = 0; k < K; k++) v[k] = q[k];
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < K; k++) s += v[k].x + v[k].w;
            acc += s;
            if (DEP) idx = mix(idx + __float_as_uint(s)) % n_records;   // the address waits for the data (a traversal step)
            else idx = mix(idx + it) % n_records;                         // independent: the loads of several rounds overlap
        }
    }
    if (acc == 12345.678f) out[tid] = acc;
}

template <int K, bool DEP>
static double run(const float4 *table, unsigned n_records, int iters, int lanes, float *out) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k_gather<K, DEP>), dim3(1024), dim3(256), 0, 0, table, n_records, iters, lanes, out);  // (warm-up: same length, so that per-dispatch counter means are of one shape)
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_gather<K, DEP>), dim3(1024), dim3(256), 0, 0, table, n_records, iters, lanes, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main(int argc, char **argv) {
    const size_t max_bytes = 64u << 20;
    std::vector<float> h(max_bytes / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) & 0xffff) * 1e-3f;
    float4 *table; float *out;
    CK(hipMalloc(&table, max_bytes));
    CK(hipMalloc(&out, 1024 * 256 * 4));
    CK(hipMemcpy(table, h.data(), max_bytes, hipMemcpyHostToDevice));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs; 1024 workgroups x 256 threads (16 waves per CU); cycles at 2.4 GHz\n", p.name, cus);
    printf("# K dep lanes table_MB   ms   wave-instr/us/CU   cycles/wave-instr/CU   GB/s(lane bytes)\n");
    const int iters = 2000;
#define ROW(K, DEP) { double ms = run<K, DEP>(table, n, iters, lanes, out); \
      double winstr = 4096.0 * iters * K; double per_us_cu = winstr / (ms * 1e3) / cus; \
              printf("%d %d %2d %3zu  %8.3f  %8.3f  %8.2f  %8.1f\n", K, (int)DEP, lanes, T >> 20, ms, per_us_cu, 2400.0 / per_us_cu, \
                     winstr * lanes * 16 / (ms * 1e-3) / 1e9); fflush(stdout); }

    if (argc == 3) {  // one configuration (for counter passes): gather_rate <lanes> <table MB>; K = 8 and K = 3, independent
        const int lanes = atoi(argv[1]);
        const size_t T = (size_t)atoi(argv[2]) << 20;
        if (lanes < 1 || lanes > 64 || T < (1u << 20) || T > max_bytes) { fprintf(stderr, "bad arguments\n"); return 2; }
        const unsigned n = (unsigned)(T / 128);
        ROW(8, false) ROW(3, false)
        return 0;
    }
    const size_t sizes[] = {1u << 20, 8u << 20, 64u << 20};
    for (size_t T : sizes) {
        const unsigned n = (unsigned)(T / 128);
        for (int lanes : {64, 40, 16}) {
            ROW(8, false) ROW(8, true) ROW(3, false) ROW(3, true) ROW(1, false) ROW(1, true)
        }
    }
    return 0;
}
