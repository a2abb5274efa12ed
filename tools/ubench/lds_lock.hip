// microbenchmark: dense row accumulation in LDS on gfx950 — float atomics vs bit-lock + plain read-modify-write
// rows: 16384 x float2 (128 KiB) + 2 KiB of lock bits; 1024 threads; every lane updates one random row per trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int kRows = 16384;
template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint32_t *locks = reinterpret_cast<uint32_t *>(lds + kRows * 2);
    for (int i = threadIdx.x; i < kRows * 2; i += 1024) lds[i] = 0.f;
    for (int i = threadIdx.x; i < kRows / 32; i += 1024) locks[i] = 0u;
    __syncthreads();
    unsigned h = (threadIdx.x + blockIdx.x * 1024) * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        const unsigned idx = (h >> 8) & (kRows - 1);
        const unsigned idx2 = idx ^ 1u;  // the x-neighbour of a hashed level differs in the low bits only
        const float a = 1.0f, b = 2.0f;
        if (MODE == 0) {  // 2 float atomics per row, one row per trip
            __hip_atomic_fetch_add(&lds[2 * idx], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lds[2 * idx + 1], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 1) {  // lock, one row per trip
            const uint32_t bit = 1u << (idx & 31u);
            uint32_t *lk = locks + (idx >> 5);
            bool pending = true;
            // wave-uniform loop condition: the holder's update + release stay INSIDE the loop (a per-lane `while (!done)` lets
            // the compiler sink them past the exit, where the winner waits for spinning losers of its own wave: deadlock)
            while (__ballot(pending)) {
                uint32_t prev = bit;
                if (pending) prev = __hip_atomic_fetch_or(lk, bit, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!(prev & bit)) {
                    float2 *cell = reinterpret_cast<float2 *>(lds) + idx;
                    float2 v = *cell;
                    v.x += a; v.y += b;
                    *cell = v;
                    __hip_atomic_fetch_and(lk, ~bit, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pending = false;
                }
            }
        } else if (MODE == 2) {  // 4 float atomics: a pair of rows per trip
            __hip_atomic_fetch_add(&lds[2 * idx], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lds[2 * idx + 1], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lds[2 * idx2], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&lds[2 * idx2 + 1], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 3) {  // one lock (on the even row) guards the aligned PAIR of rows: 16-byte read-modify-write
            const unsigned pr = idx >> 1;
            const uint32_t bit = 1u << (pr & 31u);
            uint32_t *lk = locks + (pr >> 5);
            bool pending = true;
            // wave-uniform loop condition: the holder's update + release stay INSIDE the loop (a per-lane `while (!done)` lets
            // the compiler sink them past the exit, where the winner waits for spinning losers of its own wave: deadlock)
            while (__ballot(pending)) {
                uint32_t prev = bit;
                if (pending) prev = __hip_atomic_fetch_or(lk, bit, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!(prev & bit)) {
                    float4 *cell = reinterpret_cast<float4 *>(lds) + pr;
                    float4 v = *cell;
                    v.x += a; v.y += b; v.z += a; v.w += b;
                    *cell = v;
                    __hip_atomic_fetch_and(lk, ~bit, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pending = false;
                }
            }
        } else if (MODE == 4) {  // uint atomics, 2 per row (what a fixed-point accumulator would cost)
            atomicAdd(reinterpret_cast<unsigned *>(&lds[2 * idx]), 1u);
            atomicAdd(reinterpret_cast<unsigned *>(&lds[2 * idx + 1]), 2u);
        } else if (MODE == 5) {  // 64-bit uint atomic, 1 per row
            atomicAdd(reinterpret_cast<unsigned long long *>(&lds[2 * idx]), 1ull);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + lds[777];
}
template <int MODE>
void run(const char *name, float *d, int rows_per_trip) {
    int iters = 2048;
    const int lds = kRows * 8 + kRows / 8;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 1024, lds>>>(d, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<256, 1024, lds>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double rows = 256.0 * 1024 * iters * rows_per_trip;
    printf("%-44s %8.3f ms  %7.2f G row-updates/s chip  %6.1f cycles per wave-trip per CU\n", name, ms, rows / ms * 1e-6,
           ms * 1e-3 * 2.4e9 / (16.0 * iters));
}
int main() {
    float *d; hipMalloc(&d, 4096);
    run<0>("float atomics, 1 row (2 ds_add_f32)", d, 1);
    run<1>("bit lock + float2 RMW, 1 row", d, 1);
    run<2>("float atomics, row pair (4 ds_add_f32)", d, 2);
    run<3>("bit lock + float4 RMW, aligned row pair", d, 2);
    run<4>("uint atomics, 1 row (2 ds_add_u32)", d, 1);
    run<5>("uint64 atomic, 1 row (1 ds_add_u64)", d, 1);
    return 0;
}
