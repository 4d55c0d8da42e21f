// Issue-rate microbenchmark for the VALU instructions the path tracer leans on (gfx950).
// Each kernel runs N dependent-free copies of one instruction per loop iteration on every lane,
// with enough waves per SIMD that issue, not latency, bounds it.  Prints wave-instructions per
// cycle per SIMD equivalents as "lane-ops/s" and the ratio to v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, float seed, int iters) {
    float a[UNROLL];
    v2f p[UNROLL];
    double dd[UNROLL];
    unsigned u[UNROLL];
    unsigned long long w[UNROLL];
    for (int i = 0; i < UNROLL; ++i) {
        a[i] = seed + i + threadIdx.x * 1e-3f;
        p[i] = v2f{a[i], a[i] * 0.5f};
        dd[i] = a[i];
        u[i] = (unsigned)(a[i] * 977.0f) | 1u;
        w[i] = u[i];
    }
    float s = seed * 0.999f, t = seed * 1e-3f;
    v2f ps = {s, s}, pt = {t, t};
    double ds = s, dt = t;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == 0) a[i] = __builtin_fmaf(a[i], s, t);
            if (OP == 1) p[i] = __builtin_elementwise_fma(p[i], ps, pt);
            if (OP == 2) u[i] = u[i] * 0xD2511F53u + 1u;                       // v_mul_lo_u32 (+add)
            if (OP == 3) u[i] = __umulhi(u[i], 0xD2511F53u) ^ (unsigned)it;    // v_mul_hi_u32
            if (OP == 4) w[i] = (unsigned long long)(unsigned)w[i] * 0xCD9E8D57u + (w[i] >> 32); // v_mad_u64_u32
            if (OP == 5) a[i] = __builtin_amdgcn_rcpf(a[i]);
            if (OP == 6) a[i] = __builtin_amdgcn_sqrtf(a[i]);
            if (OP == 7) dd[i] = __builtin_fma(dd[i], ds, dt);
            if (OP == 8) a[i] = __builtin_fminf(__builtin_fminf(a[i], s), t) + 1.0f;  // min3 + add
            if (OP == 9) u[i] = (u[i] & 0xFFFFFFu) * 0x51F53u + 1u;  // v_mul_u32_u24 / v_mad_u32_u24
            if (OP == 10) a[i] = a[i] * s;                                       // v_mul_f32
            if (OP == 11) a[i] = __builtin_amdgcn_rsqf(a[i]);
        }
    }
    float r = 0;
    for (int i = 0; i < UNROLL; ++i) r += a[i] + p[i].x + p[i].y + (float)dd[i] + (float)u[i] + (float)w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
double run(const char* name, double lane_ops_per_inst, float* out, int blocks) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 64);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, ITERS);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double insts = (double)blocks * 256 * (double)ITERS * UNROLL; // lane-instructions
    double rate = insts / (ms * 1e-3);
    printf("%-28s %8.3f ms  %10.3f G lane-inst/s  (%.2f T ops/s)\n", name, ms, rate * 1e-9, rate * lane_ops_per_inst * 1e-12);
    return rate;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    int blocks = prop.multiProcessorCount * 8; // 8 blocks x 4 waves = 32 waves/CU = 8 per SIMD
    float* out; CHECK(hipMalloc(&out, sizeof(float) * blocks * 256));
    double f = run<0>("v_fma_f32", 2, out, blocks);
    double r;
    r = run<1>("v_pk_fma_f32", 4, out, blocks); printf("   ratio to fma: %.3f\n", r / f);
    r = run<10>("v_mul_f32", 1, out, blocks); printf("   ratio to fma: %.3f\n", r / f);
    r = run<2>("v_mul_lo_u32(+add)", 1, out, blocks); printf("   ratio to fma: %.3f\n", r / f);
    r = run<3>("v_mul_hi_u32(+xor)", 1, out, blocks); printf("   ratio to fma: %.3f\n", r / f);
    r = run<4>("v_mad_u64_u32", 1, out, blocks); printf("   ratio to fma: %.3f\n", r / f);
    r = run<9>("v_mad_u32_u24", 1, out, blocks); printf("   ratio to fma: %.3f\n", r / f);
    r = run<5>("v_rcp_f32", 1, out, blocks); printf("   ratio to fma: %.3f\n", r / f);
    r = run<11>("v_rsq_f32", 1, out, blocks); printf("   ratio to fma: %.3f\n", r / f);
    r = run<6>("v_sqrt_f32", 1, out, blocks); printf("   ratio to fma: %.3f\n", r / f);
    r = run<7>("v_fma_f64", 2, out, blocks); printf("   ratio to fma: %.3f\n", r / f);
    r = run<8>("v_min3_f32+v_add_f32", 1, out, blocks); printf("   ratio to fma (2 inst): %.3f\n", r / f);
    return 0;
}
