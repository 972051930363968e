// Latency of a dependent chain of scalar loads of a 128-byte record from a small table (the scene
// walk's record fetch): every wavefront chases `skip` links through 48 records, 64-byte or 128-byte
// loads, alone on its CU or with 19 others.
// build: hipcc --offload-arch=gfx950 -O3 -o smem_latency smem_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define RT_CONST __attribute__((address_space(4)))
struct alignas(16) Rec { int kind, skip, ctx, node; double r0, r1; double p[12]; };

template <int BYTES>
__global__ void chase(const RT_CONST Rec* recs, int steps, long long* cycles, double* sink) {
    int i = 0;
    double acc = 0.0;
    long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        i = __builtin_amdgcn_readfirstlane(i);
        const RT_CONST Rec* r = recs + i;
        const int skip = r->skip;
        double a = r->r0, b = r->p[0], c = r->p[3];
        if (BYTES == 128) { a += r->p[6]; b += r->p[9]; c += r->p[11]; }
        asm volatile("" ::"s"(skip), "s"(a), "s"(b), "s"(c));
        acc += a + b + c;
        i = skip;
    }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    const int n = 48, steps = 4800;
    std::vector<Rec> h(n);
    for (int k = 0; k < n; ++k) { h[k] = Rec{}; h[k].skip = (k * 7 + 5) % n; h[k].r0 = k; }
    Rec* d; hipMalloc(&d, n * sizeof(Rec)); hipMemcpy(d, h.data(), n * sizeof(Rec), hipMemcpyHostToDevice);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    long long* cyc; hipMalloc(&cyc, 8 * cus * 32); double* sink; hipMalloc(&sink, 8 * cus * 32 * 64);
    for (int waves_per_cu : {1, 20}) {
        const int blocks = cus * waves_per_cu;
        for (int bytes : {64, 128}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (bytes == 64) chase<64><<<blocks, 64>>>((const RT_CONST Rec*)d, steps, cyc, sink);
                else chase<128><<<blocks, 64>>>((const RT_CONST Rec*)d, steps, cyc, sink);
                hipDeviceSynchronize();
            }
            std::vector<long long> c(blocks);
            hipMemcpy(c.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
            double sum = 0; for (auto v : c) sum += v;
            printf("%2d wavefront(s) per CU, %3d-byte records: %.1f shader-clock ticks per dependent fetch (s_memrealtime/readcyclecounter units)\n",
                   waves_per_cu, bytes, sum / blocks / steps);
        }
    }
    return 0;
}
