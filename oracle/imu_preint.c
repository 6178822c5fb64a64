/* Oracle (TEST INFRASTRUCTURE ONLY -- never linked into the product library).
 *
 * Plain-C restatement of the IMU pre-integration the reference runs per frame interval:
 *   reference imu_integrator.py:69-164   IMUModule.integrate (frame loop, motion / world mode,
 *                                         empty-interval rule :134-140, state chaining :154-157)
 *   reference imu_integrator.py:11-28    prase_init
 *   PyPose pp.module.IMUPreintegrator.forward -> integrate + predict (NOT in /root/reference,
 *   unpinned: "parity unpinned"; restated from its published algorithm):
 *       dr = [I, Exp(gyro*dt)] ; incre_r = cumprod(dr) by Hillis-Steele doubling scan
 *       a  = acc - (R0*incre_r[1:])^-1 g ; dv = [0, incre_r[:F] a dt] ; incre_v = cumsum
 *       dp = [0, incre_v[:F] dt + incre_r[:F] a 0.5 dt^2] ; incre_p = cumsum ; incre_t = cumsum(dt)
 *       rot = R0*incre_r ; vel = v0 + R0 incre_v ; pos = p0 + R0 incre_p + v0 incre_t
 *
 * Floating-point contract (what "bit-matching" in tests/test_imu_gpu.py means): every operation
 * below is a single IEEE-754 operation in the working precision, evaluated in the order written,
 * no fused multiply-add (build with -ffp-contract=off), sqrt and divide correctly rounded, and
 * sin/cos from the fdlibm kernel polynomials k_sin/k_cos evaluated in double (arguments are
 * |theta/2| << pi/4 for any physical gyro rate; a 2-term Cody-Waite reduction covers the rest).
 * The HIP kernel in islam_amd/csrc/imu_preint.hip is written independently to the same contract.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- sin/cos: fdlibm k_sin.c / k_cos.c polynomials (Sun Microsystems, freely distributable) ---- */
static double ksin(double x) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x + v * (S1 + z * r);
}
static double kcos(double x) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double ax = fabs(x);
    if (ax < 0.3) return 1.0 - (0.5 * z - z * r);
    double qx = (ax > 0.78125) ? 0.28125 : 0.25 * ax;
    double hz = 0.5 * z - qx;
    double a = 1.0 - qx;
    return a - (hz - z * r);
}
void islam_oracle_sincos(double x, double* s, double* c) {
    const double pio4 = 7.85398163397448278999e-01, invpio2 = 6.36619772367581382433e-01,
                 pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    double y = x;
    int q = 0;
    if (fabs(x) > pio4) {
        double fn = rint(x * invpio2);
        y = (x - fn * pio2_1) - fn * pio2_1t;
        q = (int)((long long)fn & 3);
    }
    double sy = ksin(y), cy = kcos(y);
    switch (q) {
        case 0: *s = sy; *c = cy; break;
        case 1: *s = cy; *c = -sy; break;
        case 2: *s = -sy; *c = -cy; break;
        default: *s = -cy; *c = sy; break;
    }
}

#define REAL double
#define SUF(n) n##_f64
#define EPS 2.220446049250313e-16
#include "imu_preint_body.inc"
#undef REAL
#undef SUF
#undef EPS

#define REAL float
#define SUF(n) n##_f32
#define EPS 1.1920928955078125e-07f
#include "imu_preint_body.inc"
#undef REAL
#undef SUF
#undef EPS
