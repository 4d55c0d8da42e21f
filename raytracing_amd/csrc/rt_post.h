// Post-processing chain on the device, fused: exposure -> contrast/brightness -> saturation ->
// global Reinhard -> gamma (upstream: src/post_processing/post_processing.py:35-82), straight from
// the accumulation buffer.
//
// Upstream's chain is a dozen whole-image numpy passes with two global reductions in the tone
// mapper (mean of log luminance, :67; maximum of the mapped luminance, :72).  The mapped luminance
// Ld = Lm (1 + Lm / s^2) / (1 + Lm), Lm = a / Lwa * Lw, grows with Lw, so its maximum is Ld at the
// maximum of Lw: BOTH reductions come out of one pass over the image (k_post_measure: per-block sum
// of log Lw in double precision and maximum of Lw, folded by k_post_fold in block order, no
// floating-point atomics), and one more pass maps and writes (k_post_apply).  HBM traffic: two reads
// and one write of the image, 36 bytes per pixel.
//
// Quirks kept because they define the output: the clip to [0, 1] happens BEFORE tone mapping (:56),
// luminance weights are (0.299, 0.587, 0.114) (:44), gamma is applied as img ** gamma (:82).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rt {

struct PostArgs {
    const float *src;      // (W, H, 3) float32, x-major
    float *dst;            // (W, H, 3) float32, x-major
    int32_t width, height;
    int32_t flip_y;        // 1: dst[x, y] takes src[x, H-1-y] (RenderResult.img_arr = arr[:, ::-1], render_result.py:46)
    float scale;           // src is multiplied by this first (1 / rendered_passes)
    float exposure, brightness, contrast, saturation, gamma, key, tone_saturation;
    double *partials;      // [2 * n_blocks + 2]: per block (sum log Lw, max Lw), then the folded pair
    int32_t n_blocks;
};

struct PostPixel { float r, g, b, lw; };

// Stages 1-3 and the tone mapper's luminance for source pixel i (post_processing.py:47-57, 62-63).
__device__ __forceinline__ PostPixel post_pixel(const PostArgs &A, size_t i) {
    const float *p = A.src + 3 * i;
    float c[3] = {p[0] * A.scale, p[1] * A.scale, p[2] * A.scale};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v = c[k] * A.exposure;
        v = A.contrast * (v - 0.5f) + A.brightness + 0.5f;
        c[k] = fminf(fmaxf(v, 0.0f), 1.0f);
    }
    const float gray = c[0] * 0.299f + c[1] * 0.587f + c[2] * 0.114f;
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] = fminf(fmaxf(gray + (c[k] - gray) * A.saturation, 0.0f), 1.0f);
    PostPixel q;
    q.r = c[0] + 1e-8f; q.g = c[1] + 1e-8f; q.b = c[2] + 1e-8f;
    q.lw = q.r * 0.299f + q.g * 0.587f + q.b * 0.114f;
    return q;
}

__global__ __launch_bounds__(256) void k_post_measure(const PostArgs A) {
    const size_t n = (size_t)A.width * A.height;
    double sum = 0.0;
    float mx = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const PostPixel q = post_pixel(A, i);
        sum += (double)logf(q.lw);
        mx = fmaxf(mx, q.lw);
    }
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_down(sum, off, 64);
        mx = fmaxf(mx, __shfl_down(mx, off, 64));
    }
    __shared__ double s_sum[4];
    __shared__ float s_max[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[wave] = sum; s_max[wave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        A.partials[2 * blockIdx.x] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
        A.partials[2 * blockIdx.x + 1] = (double)fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    }
}

__global__ __launch_bounds__(64) void k_post_fold(const PostArgs A) {
    double sum = 0.0, mx = 0.0;
    for (int b = threadIdx.x; b < A.n_blocks; b += 64) { sum += A.partials[2 * b]; mx = fmax(mx, A.partials[2 * b + 1]); }
    for (int off = 32; off > 0; off >>= 1) { sum += __shfl_down(sum, off, 64); mx = fmax(mx, __shfl_down(mx, off, 64)); }
    if (threadIdx.x == 0) { A.partials[2 * A.n_blocks] = sum; A.partials[2 * A.n_blocks + 1] = mx; }
}

__global__ __launch_bounds__(256) void k_post_apply(const PostArgs A) {
    const size_t n = (size_t)A.width * A.height;
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;   // destination pixel
    if (j >= n) return;
    const size_t x = j / (size_t)A.height, y = j - x * (size_t)A.height;
    const size_t i = A.flip_y ? x * (size_t)A.height + ((size_t)A.height - 1 - y) : j;
    const double log_sum = A.partials[2 * A.n_blocks], lw_max = A.partials[2 * A.n_blocks + 1];
    const float lwa = (float)exp(log_sum / (double)n);                // post_processing.py:67
    const float k = A.key / lwa;
    const float inv_s2 = 1.0f / (A.tone_saturation * A.tone_saturation);
    const float lm_max = k * (float)lw_max;
    const float ld_max = lm_max * (1.0f + lm_max * inv_s2) / (1.0f + lm_max);   // :69-72, at the brightest pixel
    const PostPixel q = post_pixel(A, i);
    const float lm = k * q.lw;
    const float ld = lm * (1.0f + lm * inv_s2) / (1.0f + lm) / ld_max;
    const float f = ld / q.lw;
    float c[3] = {q.r * f, q.g * f, q.b * f};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        float v = fminf(fmaxf(c[t], 0.0f), 1.0f);
        c[t] = A.gamma == 1.0f ? v : powf(v, A.gamma);              // :82
    }
    float *d = A.dst + 3 * j;
    d[0] = c[0]; d[1] = c[1]; d[2] = c[2];
}

}  // namespace rt
