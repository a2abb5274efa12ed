// microbenchmark: LDS atomic add throughput on gfx950 (float vs uint, active lanes, address pattern)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE, int ACTIVE>
__global__ void __launch_bounds__(1024) k(float *out, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 32768; i += 1024) lds[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned h = threadIdx.x * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        unsigned idx;
        if (MODE == 0 || MODE == 2) idx = (h >> 8) & 32767;       // random
        else idx = threadIdx.x;                                    // linear, conflict free
        if (lane < ACTIVE) {
            if (MODE == 2 || MODE == 3) atomicAdd((unsigned *)&lds[idx], 1u);
            else __hip_atomic_fetch_add(&lds[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + lds[777];
}
template <int MODE, int ACTIVE>
void run(const char *name, float *d) {
    int iters = 4096;
    hipFuncSetAttribute((const void *)k<MODE, ACTIVE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, ACTIVE><<<256, 1024, 131072>>>(d, 16);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE, ACTIVE><<<256, 1024, 131072>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = 256.0 * 16 * ACTIVE * iters;   // lane-atomics
    printf("%-34s active %2d: %8.3f ms  %7.2f G lane-atomics/s chip  %6.2f cycles/wave-instr/CU(@2.4GHz)\n", name, ACTIVE, ms,
           ops / ms * 1e-6, ms * 1e-3 * 2.4e9 / (16.0 * iters));
}
int main() {
    float *d; hipMalloc(&d, 4096);
    run<0, 64>("float random", d); run<0, 16>("float random", d); run<0, 2>("float random", d);
    run<1, 64>("float linear", d); run<1, 2>("float linear", d);
    run<2, 64>("uint random", d); run<2, 2>("uint random", d);
    run<3, 64>("uint linear", d);
    return 0;
}
