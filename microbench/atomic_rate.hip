// How fast does gfx950 serve atomics from many waves?  One lane per wave issues K atomic adds (developer tool):
// same address or one 128-byte line per wave, with the old value returned (the wave waits for each) or not.
// A persistent render kernel takes its jobs from one global counter: this is what that counter can deliver.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <bool RETURNING, bool SAME>
__global__ __launch_bounds__(256) void k(unsigned *ctr, int per_wave, unsigned *sink) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned *p = SAME ? ctr : ctr + 32 * (size_t)wave;
    unsigned acc = 0;
    if ((threadIdx.x & 63u) == 0) {
        for (int i = 0; i < per_wave; ++i) {
            if (RETURNING) acc += atomicAdd(p, 1u);
            else atomicAdd(p, 1u);
        }
        if (acc == 0xFFFFFFFFu) *sink = acc;
    }
}
template <bool RETURNING, bool SAME>
void run(const char *what, unsigned *ctr, unsigned *sink, int blocks, int per_wave) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemset(ctr, 0, 4);
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL((k<RETURNING, SAME>), dim3(blocks), dim3(256), 0, 0, ctr, per_wave, sink);
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double n = (double)blocks * 4 * per_wave;
    printf("%-44s %5d blocks x 4 waves x %4d: %8.3f ms = %7.1f ns per atomic, %7.1f M atomics/s\n", what, blocks, per_wave, best, best * 1e6 / n, n / best / 1e3);
}
int main() {
    unsigned *ctr, *sink;
    (void)hipMalloc(&ctr, 128 * 8192 + 128); (void)hipMalloc(&sink, 4);
    (void)hipMemset(ctr, 0, 128 * 8192 + 128);
    for (int blocks : {256, 1280}) {
        run<true, true>("same address, value returned", ctr, sink, blocks, 64);
        run<false, true>("same address, nothing returned", ctr, sink, blocks, 64);
        run<true, false>("a line per wave, value returned", ctr, sink, blocks, 64);
        run<false, false>("a line per wave, nothing returned", ctr, sink, blocks, 64);
    }
    return 0;
}
