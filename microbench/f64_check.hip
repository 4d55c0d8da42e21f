// Does double-precision device arithmetic round like the host's?  One sphere root, step by step (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *in, double *out) {
    double ox = in[0], oy = in[1], oz = in[2], dx = in[3], dy = in[4], dz = in[5], cx = in[6], cy = in[7], cz = in[8], r = in[9];
    double rx = ox - cx, ry = oy - cy, rz = oz - cz;
    double a = dx * dx + dy * dy + dz * dz;
    double b = 2 * (rx * dx + ry * dy + rz * dz);
    double c = rx * rx + ry * ry + rz * rz - r * r;
    double rad = b * b - 4 * a * c;
    double sq = __builtin_sqrt(rad);
    double t1 = (-b - sq) / (2 * a);
    out[0] = a; out[1] = b; out[2] = c; out[3] = rad; out[4] = sq; out[5] = t1; out[6] = 1.0 / a; out[7] = __builtin_sqrt(in[10]);
}
int main() {
    double in[11] = {1.375, -0.375, 0.25, -0.8944271802902222, 0.4472135901451111, 0.0, 0.5, 0.0, 0.25, 0.125, 0.049999998802620294};
    double *din, *dout, out[8], ref[8];
    hipMalloc(&din, sizeof in); hipMalloc(&dout, sizeof out);
    hipMemcpy(din, in, sizeof in, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, din, dout);
    hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
    volatile double ox = in[0], oy = in[1], oz = in[2], dx = in[3], dy = in[4], dz = in[5], cx = in[6], cy = in[7], cz = in[8], r = in[9];
    double rx = ox - cx, ry = oy - cy, rz = oz - cz;
    ref[0] = dx * dx + dy * dy + dz * dz; ref[1] = 2 * (rx * dx + ry * dy + rz * dz); ref[2] = rx * rx + ry * ry + rz * rz - r * r;
    ref[3] = ref[1] * ref[1] - 4 * ref[0] * ref[2]; ref[4] = sqrt(ref[3]); ref[5] = (-ref[1] - ref[4]) / (2 * ref[0]); ref[6] = 1.0 / ref[0]; ref[7] = sqrt(in[10]);
    const char *nm[8] = {"a", "b", "c", "rad", "sqrt", "t1", "1/a", "sqrt(const)"};
    for (int i = 0; i < 8; ++i) printf("%-12s gpu %.17g host %.17g %s\n", nm[i], out[i], ref[i], out[i] == ref[i] ? "" : "DIFFERENT");
    return 0;
}
