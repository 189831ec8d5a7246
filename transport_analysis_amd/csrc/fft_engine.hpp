// fft_engine.hpp — complex-f64 building blocks of the FFT kernels (gfx950): the element type,
// radix-2/4/5/8/16 butterflies on registers, wave-uniform twiddle loads.
#pragma once
#include <hip/hip_runtime.h>

namespace ta {

struct cd {
    double x, y;
};

__device__ __forceinline__ cd operator+(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cd operator-(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cd cmul(cd a, cd b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__device__ __forceinline__ cd cmulc(cd a, cd b) {  // a * conj(b)
    return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}
__device__ __forceinline__ cd mul_mi(cd a) { return {a.y, -a.x}; }  // a * (-i)
__device__ __forceinline__ cd mul_pi(cd a) { return {-a.y, a.x}; }  // a * (+i)
__device__ __forceinline__ double norm2(cd a) { return a.x * a.x + a.y * a.y; }

constexpr double kR2 = 0.70710678118654752440084436210485;   // sqrt(1/2)
constexpr double kC8 = 0.92387953251128675612818318939679;   // cos(pi/8)
constexpr double kS8 = 0.38268343236508977172845998403040;   // sin(pi/8)
constexpr double kC5a = 0.30901699437494742410229341718282;  // cos(2pi/5)
constexpr double kC5b = -0.80901699437494742410229341718282; // cos(4pi/5)
constexpr double kS5a = 0.95105651629515357211643933337938;  // sin(2pi/5)
constexpr double kS5b = 0.58778525229247312916870595463907;  // sin(4pi/5)

// ---- forward DFTs on registers: v[q] <- sum_j v[j] exp(-2 pi i j q / R) -------
template <int R>
struct Dft;

template <>
struct Dft<2> {
    static __device__ __forceinline__ void run(cd (&v)[2]) {
        cd a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    }
};

template <>
struct Dft<4> {
    static __device__ __forceinline__ void run(cd (&v)[4]) {
        cd t0 = v[0] + v[2], t1 = v[0] - v[2];
        cd t2 = v[1] + v[3], t3 = mul_mi(v[1] - v[3]);
        v[0] = t0 + t2;
        v[1] = t1 + t3;
        v[2] = t0 - t2;
        v[3] = t1 - t3;
    }
};

template <>
struct Dft<5> {
    static __device__ __forceinline__ void run(cd (&v)[5]) {
        cd t1 = v[1] + v[4], t2 = v[2] + v[3], t3 = v[1] - v[4], t4 = v[2] - v[3];
        cd y0 = v[0] + t1 + t2;
        cd a1 = {v[0].x + kC5a * t1.x + kC5b * t2.x, v[0].y + kC5a * t1.y + kC5b * t2.y};
        cd a2 = {v[0].x + kC5b * t1.x + kC5a * t2.x, v[0].y + kC5b * t1.y + kC5a * t2.y};
        cd b1 = {kS5a * t3.x + kS5b * t4.x, kS5a * t3.y + kS5b * t4.y};
        cd b2 = {kS5b * t3.x - kS5a * t4.x, kS5b * t3.y - kS5a * t4.y};
        cd ib1 = mul_mi(b1), ib2 = mul_mi(b2);  // -i*b
        v[0] = y0;
        v[1] = a1 + ib1;
        v[4] = a1 - ib1;
        v[2] = a2 + ib2;
        v[3] = a2 - ib2;
    }
};

template <>
struct Dft<8> {
    static __device__ __forceinline__ void run(cd (&v)[8]) {
        cd e[4] = {v[0], v[2], v[4], v[6]};
        cd o[4] = {v[1], v[3], v[5], v[7]};
        Dft<4>::run(e);
        Dft<4>::run(o);
        cd o1 = {kR2 * (o[1].x + o[1].y), kR2 * (o[1].y - o[1].x)};   // * (1-i)/sqrt2
        cd o2 = mul_mi(o[2]);
        cd o3 = {kR2 * (o[3].y - o[3].x), -kR2 * (o[3].x + o[3].y)};  // * (-1-i)/sqrt2
        v[0] = e[0] + o[0];
        v[4] = e[0] - o[0];
        v[1] = e[1] + o1;
        v[5] = e[1] - o1;
        v[2] = e[2] + o2;
        v[6] = e[2] - o2;
        v[3] = e[3] + o3;
        v[7] = e[3] - o3;
    }
};

// multiply by W16^n = exp(-2 pi i n / 16) for the n that occur in a 4x4 split
template <int N>
__device__ __forceinline__ cd mul_w16(cd a) {
    if constexpr (N == 0) return a;
    else if constexpr (N == 1) return {kC8 * a.x + kS8 * a.y, kC8 * a.y - kS8 * a.x};
    else if constexpr (N == 2) return {kR2 * (a.x + a.y), kR2 * (a.y - a.x)};
    else if constexpr (N == 3) return {kS8 * a.x + kC8 * a.y, kS8 * a.y - kC8 * a.x};
    else if constexpr (N == 4) return mul_mi(a);
    else if constexpr (N == 6) return {kR2 * (a.y - a.x), -kR2 * (a.x + a.y)};
    else if constexpr (N == 9) return {-kC8 * a.x - kS8 * a.y, kS8 * a.x - kC8 * a.y};
    else return a;
}

template <>
struct Dft<16> {
    static __device__ __forceinline__ void run(cd (&v)[16]) {
        // j = 4*j1 + j2 ; q = q1 + 4*q2
        cd c0[4] = {v[0], v[4], v[8], v[12]};
        cd c1[4] = {v[1], v[5], v[9], v[13]};
        cd c2[4] = {v[2], v[6], v[10], v[14]};
        cd c3[4] = {v[3], v[7], v[11], v[15]};
        Dft<4>::run(c0);
        Dft<4>::run(c1);
        Dft<4>::run(c2);
        Dft<4>::run(c3);
        // twiddle W16^(j2*q1), then DFT4 over j2 for each q1
        cd r0[4] = {c0[0], c1[0], c2[0], c3[0]};
        cd r1[4] = {c0[1], mul_w16<1>(c1[1]), mul_w16<2>(c2[1]), mul_w16<3>(c3[1])};
        cd r2[4] = {c0[2], mul_w16<2>(c1[2]), mul_w16<4>(c2[2]), mul_w16<6>(c3[2])};
        cd r3[4] = {c0[3], mul_w16<3>(c1[3]), mul_w16<6>(c2[3]), mul_w16<9>(c3[3])};
        Dft<4>::run(r0);
        Dft<4>::run(r1);
        Dft<4>::run(r2);
        Dft<4>::run(r3);
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) {
            v[0 + 4 * q2] = r0[q2];
            v[1 + 4 * q2] = r1[q2];
            v[2 + 4 * q2] = r2[q2];
            v[3 + 4 * q2] = r3[q2];
        }
    }
};

// Lane-uniform table entries go through the scalar data cache (s_load_dwordx4, constant address
// space): no vector register, no vector-memory queue.
typedef double tw_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cd tw_uniform(const cd* tw2, int idx) {
    const tw_d2 t = ((const tw_d2 __attribute__((address_space(4)))*)tw2)[idx];
    return cd{t.x, t.y};
}

}  // namespace ta
