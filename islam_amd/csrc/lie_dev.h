// SO(3)/SE(3) device helpers for the gfx950 kernels (PVGO linearisation, retraction, losses).
//
// Conventions are the ones PyPose's LieTensor uses and the reference relies on
// (reference pvgo.py:36-51, Datasets/transformation.py:72-124): quaternion [x,y,z,w],
// SE3 = [t, q], tangent [rho, phi], left perturbation Exp(d)*X.
//
// Everything is register-resident: 3x3 matrices are nine named scalars in a struct that is only
// ever indexed with compile-time constants (runtime-indexed arrays would go to scratch memory).
// Small-angle branches switch to Taylor series below 1e-2 rad: PyPose switches only at machine
// eps, where its closed forms lose up to all digits to cancellation ((t^2+2cos t-2)/2t^4 ...);
// the series agree with the closed forms to < 1e-13 at the switch point.
#pragma once
#include <hip/hip_runtime.h>

namespace islam {

#define ISLAM_DEV __device__ __forceinline__

template <class T> struct V3 { T x, y, z; };
template <class T> struct Q4 { T x, y, z, w; };
template <class T> struct M3 { T a00, a01, a02, a10, a11, a12, a20, a21, a22; };
template <class T> struct SE3 { V3<T> t; Q4<T> q; };

template <class T> ISLAM_DEV V3<T> v3(T x, T y, T z) { return V3<T>{x, y, z}; }
template <class T> ISLAM_DEV V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> ISLAM_DEV V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> ISLAM_DEV V3<T> operator-(V3<T> a) { return {-a.x, -a.y, -a.z}; }
template <class T> ISLAM_DEV V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class T> ISLAM_DEV T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> ISLAM_DEV V3<T> cross(V3<T> a, V3<T> b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

template <class T> ISLAM_DEV M3<T> m3_identity() { return {T(1), T(0), T(0), T(0), T(1), T(0), T(0), T(0), T(1)}; }
template <class T> ISLAM_DEV M3<T> skew(V3<T> v) { return {T(0), -v.z, v.y, v.z, T(0), -v.x, -v.y, v.x, T(0)}; }
template <class T> ISLAM_DEV M3<T> operator+(M3<T> a, M3<T> b) {
    return {a.a00 + b.a00, a.a01 + b.a01, a.a02 + b.a02, a.a10 + b.a10, a.a11 + b.a11, a.a12 + b.a12,
            a.a20 + b.a20, a.a21 + b.a21, a.a22 + b.a22};
}
template <class T> ISLAM_DEV M3<T> operator-(M3<T> a, M3<T> b) {
    return {a.a00 - b.a00, a.a01 - b.a01, a.a02 - b.a02, a.a10 - b.a10, a.a11 - b.a11, a.a12 - b.a12,
            a.a20 - b.a20, a.a21 - b.a21, a.a22 - b.a22};
}
template <class T> ISLAM_DEV M3<T> operator*(T s, M3<T> a) {
    return {s * a.a00, s * a.a01, s * a.a02, s * a.a10, s * a.a11, s * a.a12, s * a.a20, s * a.a21, s * a.a22};
}
template <class T> ISLAM_DEV M3<T> operator*(M3<T> a, M3<T> b) {
    return {a.a00 * b.a00 + a.a01 * b.a10 + a.a02 * b.a20, a.a00 * b.a01 + a.a01 * b.a11 + a.a02 * b.a21,
            a.a00 * b.a02 + a.a01 * b.a12 + a.a02 * b.a22,
            a.a10 * b.a00 + a.a11 * b.a10 + a.a12 * b.a20, a.a10 * b.a01 + a.a11 * b.a11 + a.a12 * b.a21,
            a.a10 * b.a02 + a.a11 * b.a12 + a.a12 * b.a22,
            a.a20 * b.a00 + a.a21 * b.a10 + a.a22 * b.a20, a.a20 * b.a01 + a.a21 * b.a11 + a.a22 * b.a21,
            a.a20 * b.a02 + a.a21 * b.a12 + a.a22 * b.a22};
}
template <class T> ISLAM_DEV M3<T> transpose(M3<T> a) {
    return {a.a00, a.a10, a.a20, a.a01, a.a11, a.a21, a.a02, a.a12, a.a22};
}
template <class T> ISLAM_DEV V3<T> operator*(M3<T> a, V3<T> v) {
    return {a.a00 * v.x + a.a01 * v.y + a.a02 * v.z, a.a10 * v.x + a.a11 * v.y + a.a12 * v.z,
            a.a20 * v.x + a.a21 * v.y + a.a22 * v.z};
}
// a^T v
template <class T> ISLAM_DEV V3<T> tmul(M3<T> a, V3<T> v) {
    return {a.a00 * v.x + a.a10 * v.y + a.a20 * v.z, a.a01 * v.x + a.a11 * v.y + a.a21 * v.z,
            a.a02 * v.x + a.a12 * v.y + a.a22 * v.z};
}
template <class T> ISLAM_DEV void m3_store(M3<T> a, T* p) {
    p[0] = a.a00; p[1] = a.a01; p[2] = a.a02; p[3] = a.a10; p[4] = a.a11; p[5] = a.a12;
    p[6] = a.a20; p[7] = a.a21; p[8] = a.a22;
}
template <class T> ISLAM_DEV M3<T> m3_load(const T* p) { return {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]}; }

// ------------------------------------------------------------------ quaternions
template <class T> ISLAM_DEV Q4<T> qmul(Q4<T> a, Q4<T> b) {
    return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
            a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
template <class T> ISLAM_DEV Q4<T> qinv(Q4<T> q) { return {-q.x, -q.y, -q.z, q.w}; }
template <class T> ISLAM_DEV V3<T> qact(Q4<T> q, V3<T> p) {
    V3<T> u{q.x, q.y, q.z};
    V3<T> uv = T(2) * cross(u, p);
    return p + q.w * uv + cross(u, uv);
}
template <class T> ISLAM_DEV M3<T> qmat(Q4<T> q) {
    T x = q.x, y = q.y, z = q.z, w = q.w;
    return {1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
            2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)};
}

template <class T> ISLAM_DEV Q4<T> so3_exp(V3<T> phi) {
    T th2 = dot(phi, phi);
    T th = sqrt(th2);
    T imag, real;
    if (th > T(1e-2)) {
        T s, c;
        sincos(T(0.5) * th, &s, &c);
        imag = s / th;
        real = c;
    } else {
        T th4 = th2 * th2;
        imag = T(0.5) - th2 * T(1.0 / 48.0) + th4 * T(1.0 / 3840.0) - th4 * th2 * T(1.0 / 645120.0);
        real = T(1.0) - th2 * T(1.0 / 8.0) + th4 * T(1.0 / 384.0) - th4 * th2 * T(1.0 / 46080.0);
    }
    return {phi.x * imag, phi.y * imag, phi.z * imag, real};
}

template <class T> ISLAM_DEV V3<T> so3_log(Q4<T> q) {
    V3<T> v{q.x, q.y, q.z};
    T vn2 = dot(v, v);
    T vn = sqrt(vn2);
    T f;
    if (vn > T(1e-3)) {
        f = T(2) * atan(vn / q.w) / vn;          // atan (not atan2), as PyPose SO3_Log
    } else {
        T w2 = q.w * q.w;                          // 2*atan(x)/vn, x = vn/w:  2/w (1 - x^2/3 + x^4/5)
        T x2 = vn2 / w2;
        f = (T(2) / q.w) * (T(1) - x2 * T(1.0 / 3.0) + x2 * x2 * T(0.2));
    }
    return f * v;
}

// Jl^-1(phi) = I - K/2 + c K^2
template <class T> ISLAM_DEV M3<T> so3_Jl_inv(V3<T> phi) {
    T th2 = dot(phi, phi);
    T c;
    if (th2 > T(1e-4)) {
        T th = sqrt(th2);
        T s, co;
        sincos(T(0.5) * th, &s, &co);
        c = (T(1) - th * co / (T(2) * s)) / th2;
    } else {
        c = T(1.0 / 12.0) + th2 * T(1.0 / 720.0) + th2 * th2 * T(1.0 / 30240.0);
    }
    M3<T> K = skew(phi);
    return m3_identity<T>() - T(0.5) * K + c * (K * K);
}

template <class T> ISLAM_DEV M3<T> so3_Jl(V3<T> phi) {
    T th2 = dot(phi, phi);
    T c1, c2;
    if (th2 > T(1e-4)) {
        T th = sqrt(th2);
        T s, co;
        sincos(th, &s, &co);
        c1 = (T(1) - co) / th2;
        c2 = (th - s) / (th2 * th);
    } else {
        c1 = T(0.5) - th2 * T(1.0 / 24.0) + th2 * th2 * T(1.0 / 720.0);
        c2 = T(1.0 / 6.0) - th2 * T(1.0 / 120.0) + th2 * th2 * T(1.0 / 5040.0);
    }
    M3<T> K = skew(phi);
    return m3_identity<T>() + c1 * K + c2 * (K * K);
}

// Barfoot's Q(rho, phi) (PyPose calcQ)
template <class T> ISLAM_DEV M3<T> se3_Q(V3<T> rho, V3<T> phi) {
    T th2 = dot(phi, phi);
    T c1, c2, c3;
    if (th2 > T(1e-4)) {
        T th = sqrt(th2);
        T s, co;
        sincos(th, &s, &co);
        T th4 = th2 * th2;
        c1 = (th - s) / (th2 * th);
        c2 = (th2 + T(2) * co - T(2)) / (T(2) * th4);
        c3 = (T(2) * th - T(3) * s + th * co) / (T(2) * th4 * th);
    } else {
        c1 = T(1.0 / 6.0) - th2 * T(1.0 / 120.0) + th2 * th2 * T(1.0 / 5040.0);
        c2 = T(1.0 / 24.0) - th2 * T(1.0 / 720.0) + th2 * th2 * T(1.0 / 40320.0);
        c3 = T(1.0 / 120.0) - th2 * T(1.0 / 2520.0) + th2 * th2 * T(1.0 / 120960.0);
    }
    M3<T> Tm = skew(rho), P = skew(phi);
    M3<T> PT = P * Tm, TP = Tm * P;
    M3<T> PTP = PT * P;
    return T(0.5) * Tm + c1 * (PT + TP + PTP) + c2 * (P * PT + TP * P - T(3) * PTP) + c3 * (PTP * P + P * PTP);
}

// ------------------------------------------------------------------ SE3
template <class T> ISLAM_DEV SE3<T> se3_load(const T* p) { return {{p[0], p[1], p[2]}, {p[3], p[4], p[5], p[6]}}; }
template <class T> ISLAM_DEV void se3_store(SE3<T> X, T* p) {
    p[0] = X.t.x; p[1] = X.t.y; p[2] = X.t.z; p[3] = X.q.x; p[4] = X.q.y; p[5] = X.q.z; p[6] = X.q.w;
}
template <class T> ISLAM_DEV SE3<T> se3_mul(SE3<T> X, SE3<T> Y) { return {X.t + qact(X.q, Y.t), qmul(X.q, Y.q)}; }
template <class T> ISLAM_DEV SE3<T> se3_inv(SE3<T> X) {
    Q4<T> qi = qinv(X.q);
    return {-qact(qi, X.t), qi};
}
template <class T> ISLAM_DEV SE3<T> se3_exp(V3<T> rho, V3<T> phi) { return {so3_Jl(phi) * rho, so3_exp(phi)}; }
template <class T> ISLAM_DEV void se3_log(SE3<T> X, V3<T>& rho, V3<T>& phi) {
    phi = so3_log(X.q);
    rho = so3_Jl_inv(phi) * X.t;
}

}  // namespace islam
