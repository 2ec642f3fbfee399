// RANSAC-EPnP pose solver (C ABI: include/pnp.h; reference call site: src/utils/eval_utils.py:18-42).
//
// fp64 throughout, like the reference's float64 cv2 call.  Latency-bound small dense algebra, not MFMA work:
//   hyp_kernel    one thread per hypothesis: hash-sampled minimal set of 5 -> EPnP -> [R | t]
//   score_kernel  one wave per hypothesis: squared reprojection error of every correspondence, ballot count
//   best_kernel   one workgroup: first arg-max of the inlier counts, inlier mask + ordered index list of the best model
//   refit_kernel  one wave: EPnP over the inliers (point sums by butterfly reductions: every lane ends with the same
//                 bits, then runs the same dense stage), pose written as [R | t / scale]
// The EPnP steps follow OpenCV calib3d/epnp.cpp (see oracle/pnp_oracle.py for the restated algorithm and citations).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <type_traits>

#include "../../include/pnp.h"

namespace pnp {

struct Cam {
    double fu, fv, uc, vc;
};
constexpr int MODEL_POINTS = 5;

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// MODEL_POINTS distinct indices in [0, n): successive hash draws, duplicates rejected (oracle: sample_indices)
__device__ void sample_indices(unsigned long long seed, int hyp, int n, int (&idx)[MODEL_POINTS]) {
    int got = 0;
    unsigned long long ctr = 0;
    while (got < MODEL_POINTS) {
        const unsigned long long r = splitmix64((seed << 40) ^ ((unsigned long long)hyp << 8) ^ ctr);
        ++ctr;
        const int v = (int)((r >> 11) % (unsigned long long)n);
        bool dup = false;
        for (int k = 0; k < got; ++k) dup |= idx[k] == v;
        if (!dup) idx[got++] = v;
    }
}

// ---- small dense algebra ------------------------------------------------------------------------------
// Everything below is written so that every array index is a compile-time constant after unrolling: the matrices then
// live in registers.  (A first version with run-time indices kept them in scratch memory and one 5-point EPnP took
// 10 ms of serial scratch latency.)

// index of element (i, j) of a symmetric N x N matrix stored as its upper triangle, row by row
template <int N>
__host__ __device__ constexpr int tri(int i, int j) {
    return i <= j ? i * N - i * (i - 1) / 2 + (j - i) : j * N - j * (j - 1) / 2 + (i - j);
}

// cyclic Jacobi for a symmetric N x N matrix held as its packed upper triangle S (N (N + 1) / 2 doubles: together with
// the N x N eigenvector matrix that is 444 registers for N = 12 instead of 576): on exit S[tri(i, i)] = eigenvalues,
// columns of V = eigenvectors.  The (p, q) sweep is fully unrolled; the sweep loop is rolled and stops when the
// off-diagonal mass is below 1e-30 of the diagonal mass.
template <int N>
__device__ __forceinline__ void jacobi_eig(double (&S)[N * (N + 1) / 2], double (&V)[N][N], int max_sweeps) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sw = 0; sw < max_sweeps; ++sw) {
        double off = 0.0, diag = 0.0;
#pragma unroll
        for (int p = 0; p < N; ++p) {
            diag += S[tri<N>(p, p)] * S[tri<N>(p, p)];
#pragma unroll
            for (int q = p + 1; q < N; ++q) off += S[tri<N>(p, q)] * S[tri<N>(p, q)];
        }
        if (!(off > 1e-30 * diag)) break;
#pragma unroll
        for (int p = 0; p < N - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                const double apq = S[tri<N>(p, q)];
                if (apq != 0.0) {
                    const double theta = (S[tri<N>(q, q)] - S[tri<N>(p, p)]) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        if (k != p && k != q) {
                            const double akp = S[tri<N>(k, p)], akq = S[tri<N>(k, q)];
                            S[tri<N>(k, p)] = c * akp - s * akq;
                            S[tri<N>(k, q)] = s * akp + c * akq;
                        }
                        const double vkp = V[k][p], vkq = V[k][q];
                        V[k][p] = c * vkp - s * vkq;
                        V[k][q] = s * vkp + c * vkq;
                    }
                    S[tri<N>(p, p)] -= t * apq;
                    S[tri<N>(q, q)] += t * apq;
                    S[tri<N>(p, q)] = 0.0;
                }
            }
    }
}

// least squares min |A x - b| for a 6 x NC system by Householder QR
template <int NC>
__device__ __forceinline__ void lstsq6(double (&A)[6][NC], double (&b)[6], double (&x)[NC]) {
#pragma unroll
    for (int k = 0; k < NC; ++k) {
        double nrm = 0.0;
#pragma unroll
        for (int i = k; i < 6; ++i) nrm += A[i][k] * A[i][k];
        nrm = sqrt(nrm);
        const double alpha = A[k][k] > 0.0 ? -nrm : nrm;
        double v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = i < k ? 0.0 : A[i][k];
        v[k] -= alpha;
        double vv = 0.0;
#pragma unroll
        for (int i = k; i < 6; ++i) vv += v[i] * v[i];
        if (vv > 0.0) {
#pragma unroll
            for (int j = k; j < NC; ++j) {
                double d = 0.0;
#pragma unroll
                for (int i = k; i < 6; ++i) d += v[i] * A[i][j];
                d = 2.0 * d / vv;
#pragma unroll
                for (int i = k; i < 6; ++i) A[i][j] -= d * v[i];
            }
            double d = 0.0;
#pragma unroll
            for (int i = k; i < 6; ++i) d += v[i] * b[i];
            d = 2.0 * d / vv;
#pragma unroll
            for (int i = k; i < 6; ++i) b[i] -= d * v[i];
        }
    }
#pragma unroll
    for (int k = NC - 1; k >= 0; --k) {
        double s = b[k];
#pragma unroll
        for (int j = k + 1; j < NC; ++j) s -= A[k][j] * x[j];
        x[k] = s / A[k][k];
    }
}

// eigen-decomposition of a symmetric 3x3 with the eigenpairs sorted by DESCENDING eigenvalue: w[k], columns E[.][k]
__device__ __forceinline__ void eig3_desc(double (&S)[6], double (&w)[3], double (&E)[3][3]) {
    double V[3][3];
    jacobi_eig<3>(S, V, 16);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        w[k] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) E[i][k] = 0.0;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int rank = 0;                           // number of eigenvalues ordered before column c (ties: lower index first)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            rank += (S[tri<3>(j, j)] > S[tri<3>(c, c)]) || (S[tri<3>(j, j)] == S[tri<3>(c, c)] && j < c);
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (rank == k) {
                w[k] = S[tri<3>(c, c)];
#pragma unroll
                for (int i = 0; i < 3; ++i) E[i][k] = V[i][c];
            }
    }
}

// R = U V^T of the SVD of a 3x3 matrix (absolute orientation), reflection fixed as epnp.cpp does (third row negated)
__device__ __forceinline__ void procrustes_rotation(const double (&M)[3][3], double (&R)[3][3]) {
    double B[6], w[3], Vs[3][3], U[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = i; j < 3; ++j) B[tri<3>(i, j)] = M[0][i] * M[0][j] + M[1][i] * M[1][j] + M[2][i] * M[2][j];   // M^T M
    eig3_desc(B, w, Vs);
    const double s0 = sqrt(fmax(w[0], 0.0));
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double sk = sqrt(fmax(w[k], 0.0));
#pragma unroll
        for (int i = 0; i < 3; ++i) U[i][k] = (M[i][0] * Vs[0][k] + M[i][1] * Vs[1][k] + M[i][2] * Vs[2][k]) / sk;
    }
    const double s2 = sqrt(fmax(w[2], 0.0));
    if (s2 > 1e-12 * s0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) U[i][2] = (M[i][0] * Vs[0][2] + M[i][1] * Vs[1][2] + M[i][2] * Vs[2][2]) / s2;
    } else {                                                   // rank 2: complete the frame
        U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
        U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
        U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = U[i][0] * Vs[j][0] + U[i][1] * Vs[j][1] + U[i][2] * Vs[j][2];
    const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                       R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
    if (det < 0.0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) R[2][j] = -R[2][j];
    }
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) x += __shfl_xor(x, o);   // butterfly: a + b == b + a, so every lane ends with the same bits
    return x;
}

// ---- EPnP ---------------------------------------------------------------------------------------------------
// NPTS > 0: the calling thread solves its own NPTS-point problem (point loops unrolled, get(i) with constant i).
// NPTS == 0: the 64 lanes of a wave share the point loops of ONE n-point problem (butterfly reductions) and all run the
// dense stage on identical data.  get(i, pw, uv): i-th correspondence (object point already scaled).
// Returns false if the pose is not finite.
template <int NPTS, class Get>
__device__ __forceinline__ bool epnp_solve(int n_rt, Get get, const Cam& cam, double (&Rout)[3][3], double (&tout)[3]) {
    constexpr bool WAVE = NPTS == 0;
    const int n = WAVE ? n_rt : NPTS;
    const int lane = WAVE ? (threadIdx.x & 63) : 0;
    auto red = [](double x) { return WAVE ? wave_sum(x) : x; };
    // for_points(f): f(i) over this lane's share of the correspondences
    auto for_points = [&](auto f) {
        if constexpr (WAVE) {
            for (int i = lane; i < n; i += 64) f(i);
        } else {
#pragma unroll
            for (int i = 0; i < NPTS; ++i) f(i);
        }
    };
    // control points: centroid + sqrt(eigenvalue / n) * principal directions (choose_control_points)
    double c0[3] = {0, 0, 0};
    for_points([&](int i) {
        double pw[3], uv[2];
        get(i, pw, uv);
        c0[0] += pw[0]; c0[1] += pw[1]; c0[2] += pw[2];
    });
#pragma unroll
    for (int k = 0; k < 3; ++k) c0[k] = red(c0[k]) / n;
    double cov[6] = {};
    for_points([&](int i) {
        double pw[3], uv[2];
        get(i, pw, uv);
        const double d[3] = {pw[0] - c0[0], pw[1] - c0[1], pw[2] - c0[2]};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = a; b < 3; ++b) cov[tri<3>(a, b)] += d[a] * d[b];
    });
#pragma unroll
    for (int k = 0; k < 6; ++k) cov[k] = red(cov[k]);
    double w3[3], EV[3][3];
    eig3_desc(cov, w3, EV);
    double cws[4][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) cws[0][k] = c0[k];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        // canonical eigenvector sign (largest-magnitude component positive, first on ties): with noisy data the solution
        // depends at noise level on which of the two mirrored control points is used
        const double a0 = fabs(EV[0][j]), a1 = fabs(EV[1][j]), a2 = fabs(EV[2][j]);
        const double lead = (a0 >= a1 && a0 >= a2) ? EV[0][j] : (a1 >= a2 ? EV[1][j] : EV[2][j]);
        const double kk = (lead < 0.0 ? -1.0 : 1.0) * sqrt(fmax(w3[j], 0.0) / n);
#pragma unroll
        for (int k = 0; k < 3; ++k) cws[j + 1][k] = c0[k] + kk * EV[k][j];
    }
    // barycentric coordinates: inverse of CC = [cws1 - cws0 | cws2 - cws0 | cws3 - cws0]
    double CC[3][3], CI[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) CC[k][j] = cws[j + 1][k] - cws[0][k];
    const double det = CC[0][0] * (CC[1][1] * CC[2][2] - CC[1][2] * CC[2][1]) - CC[0][1] * (CC[1][0] * CC[2][2] - CC[1][2] * CC[2][0]) +
                       CC[0][2] * (CC[1][0] * CC[2][1] - CC[1][1] * CC[2][0]);
    CI[0][0] = (CC[1][1] * CC[2][2] - CC[1][2] * CC[2][1]) / det; CI[0][1] = (CC[0][2] * CC[2][1] - CC[0][1] * CC[2][2]) / det;
    CI[0][2] = (CC[0][1] * CC[1][2] - CC[0][2] * CC[1][1]) / det; CI[1][0] = (CC[1][2] * CC[2][0] - CC[1][0] * CC[2][2]) / det;
    CI[1][1] = (CC[0][0] * CC[2][2] - CC[0][2] * CC[2][0]) / det; CI[1][2] = (CC[0][2] * CC[1][0] - CC[0][0] * CC[1][2]) / det;
    CI[2][0] = (CC[1][0] * CC[2][1] - CC[1][1] * CC[2][0]) / det; CI[2][1] = (CC[0][1] * CC[2][0] - CC[0][0] * CC[2][1]) / det;
    CI[2][2] = (CC[0][0] * CC[1][1] - CC[0][1] * CC[1][0]) / det;
    auto alphas = [&](const double (&pw)[3], double (&a)[4]) {
        const double d[3] = {pw[0] - c0[0], pw[1] - c0[1], pw[2] - c0[2]};
#pragma unroll
        for (int j = 0; j < 3; ++j) a[j + 1] = CI[j][0] * d[0] + CI[j][1] * d[1] + CI[j][2] * d[2];
        a[0] = 1.0 - a[1] - a[2] - a[3];
    };
    // M^T M (fill_M), mean alphas, G_j = sum_i alpha_ij (pw_i - pw0)
    double A[78] = {};                      // M^T M, packed upper triangle
    double abar[4] = {0, 0, 0, 0}, G[4][3] = {};
    for_points([&](int i) {
        double pw[3], uv[2], a[4];
        get(i, pw, uv);
        alphas(pw, a);
        double r1[12], r2[12];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r1[3 * j] = a[j] * cam.fu; r1[3 * j + 1] = 0.0; r1[3 * j + 2] = a[j] * (cam.uc - uv[0]);
            r2[3 * j] = 0.0; r2[3 * j + 1] = a[j] * cam.fv; r2[3 * j + 2] = a[j] * (cam.vc - uv[1]);
            abar[j] += a[j];
#pragma unroll
            for (int k = 0; k < 3; ++k) G[j][k] += a[j] * (pw[k] - c0[k]);
        }
#pragma unroll
        for (int p = 0; p < 12; ++p)
#pragma unroll
            for (int q = p; q < 12; ++q) A[tri<12>(p, q)] += r1[p] * r1[q] + r2[p] * r2[q];
    });
#pragma unroll
    for (int k = 0; k < 78; ++k) A[k] = red(A[k]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        abar[j] = red(abar[j]) / n;
#pragma unroll
        for (int k = 0; k < 3; ++k) G[j][k] = red(G[j][k]);
    }
    double afirst[4];
    {
        double pw[3], uv[2];
        get(0, pw, uv);
        alphas(pw, afirst);
    }
    // null space of M: the 4 eigenvectors of M^T M with the smallest eigenvalues (v[0] = smallest)
    double v[4][12];
    {
        double EV12[12][12];
        jacobi_eig<12>(A, EV12, 24);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int k = 0; k < 12; ++k) v[s][k] = 0.0;
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            int rank = 0;                       // eigenvalues ordered before column c (ascending, ties: lower index first)
#pragma unroll
            for (int j = 0; j < 12; ++j)
                rank += (A[tri<12>(j, j)] < A[tri<12>(c, c)]) || (A[tri<12>(j, j)] == A[tri<12>(c, c)] && j < c);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                if (rank == s) {
#pragma unroll
                    for (int k = 0; k < 12; ++k) v[s][k] = EV12[k][c];
                }
        }
    }
    // L (6 x 10) and rho (compute_L_6x10, compute_rho)
    constexpr int PA[6] = {0, 0, 0, 1, 1, 2}, PB[6] = {1, 2, 3, 2, 3, 3};
    double L[6][10], rho[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        double dv[4][3];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int k = 0; k < 3; ++k) dv[s][k] = v[s][3 * PA[r] + k] - v[s][3 * PB[r] + k];
        auto dot = [&](int i, int j) { return dv[i][0] * dv[j][0] + dv[i][1] * dv[j][1] + dv[i][2] * dv[j][2]; };
        L[r][0] = dot(0, 0); L[r][1] = 2 * dot(0, 1); L[r][2] = dot(1, 1); L[r][3] = 2 * dot(0, 2); L[r][4] = 2 * dot(1, 2);
        L[r][5] = dot(2, 2); L[r][6] = 2 * dot(0, 3); L[r][7] = 2 * dot(1, 3); L[r][8] = 2 * dot(2, 3); L[r][9] = dot(3, 3);
        double d2 = 0.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) d2 += (cws[PA[r]][k] - cws[PB[r]][k]) * (cws[PA[r]][k] - cws[PB[r]][k]);
        rho[r] = d2;
    }
    // three beta approximations, 5 Gauss-Newton steps each, pose of each, smallest reprojection error wins
    double Rc[3][3][3], tc[3][3];
    auto candidate = [&](auto NA, double (&Rk)[3][3], double (&tk)[3]) {
        constexpr int na = decltype(NA)::value;
        double b[4] = {0, 0, 0, 0};
        if constexpr (na == 0) {                                  // unknowns B11 B12 B13 B14
            double Aq[6][4], bq[6], x[4];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                Aq[r][0] = L[r][0]; Aq[r][1] = L[r][1]; Aq[r][2] = L[r][3]; Aq[r][3] = L[r][6];
                bq[r] = rho[r];
            }
            lstsq6<4>(Aq, bq, x);
            if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = -x[1] / b[0]; b[2] = -x[2] / b[0]; b[3] = -x[3] / b[0]; }
            else { b[0] = sqrt(x[0]); b[1] = x[1] / b[0]; b[2] = x[2] / b[0]; b[3] = x[3] / b[0]; }
        } else if constexpr (na == 1) {                           // B11 B12 B22
            double Aq[6][3], bq[6], x[3];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                Aq[r][0] = L[r][0]; Aq[r][1] = L[r][1]; Aq[r][2] = L[r][2];
                bq[r] = rho[r];
            }
            lstsq6<3>(Aq, bq, x);
            if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
            else { b[0] = sqrt(x[0]); b[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) b[0] = -b[0];
        } else {                                                  // B11 B12 B22 B13 B23
            double Aq[6][5], bq[6], x[5];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c = 0; c < 5; ++c) Aq[r][c] = L[r][c];
                bq[r] = rho[r];
            }
            lstsq6<5>(Aq, bq, x);
            if (x[0] < 0) { b[0] = sqrt(-x[0]); b[1] = x[2] < 0 ? sqrt(-x[2]) : 0.0; }
            else { b[0] = sqrt(x[0]); b[1] = x[2] > 0 ? sqrt(x[2]) : 0.0; }
            if (x[1] < 0) b[0] = -b[0];
            b[2] = x[3] / b[0];
        }
        for (int it = 0; it < 5; ++it) {                          // gauss_newton
            double Aq[6][4], bq[6], x[4];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const double* l = L[r];
                Aq[r][0] = 2 * l[0] * b[0] + l[1] * b[1] + l[3] * b[2] + l[6] * b[3];
                Aq[r][1] = l[1] * b[0] + 2 * l[2] * b[1] + l[4] * b[2] + l[7] * b[3];
                Aq[r][2] = l[3] * b[0] + l[4] * b[1] + 2 * l[5] * b[2] + l[8] * b[3];
                Aq[r][3] = l[6] * b[0] + l[7] * b[1] + l[8] * b[2] + 2 * l[9] * b[3];
                bq[r] = rho[r] - (l[0] * b[0] * b[0] + l[1] * b[0] * b[1] + l[2] * b[1] * b[1] + l[3] * b[0] * b[2] + l[4] * b[1] * b[2] +
                                  l[5] * b[2] * b[2] + l[6] * b[0] * b[3] + l[7] * b[1] * b[3] + l[8] * b[2] * b[3] + l[9] * b[3] * b[3]);
            }
            lstsq6<4>(Aq, bq, x);
#pragma unroll
            for (int k = 0; k < 4; ++k) b[k] += x[k];
        }
        // compute_ccs / solve_for_sign / estimate_R_and_t
        double ccs[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) ccs[j][k] = b[0] * v[0][3 * j + k] + b[1] * v[1][3 * j + k] + b[2] * v[2][3 * j + k] + b[3] * v[3][3 * j + k];
        const double z0 = afirst[0] * ccs[0][2] + afirst[1] * ccs[1][2] + afirst[2] * ccs[2][2] + afirst[3] * ccs[3][2];
        const double sg = z0 < 0.0 ? -1.0 : 1.0;
        double pc0[3], ABt[3][3];
#pragma unroll
        for (int k = 0; k < 3; ++k) pc0[k] = sg * (abar[0] * ccs[0][k] + abar[1] * ccs[1][k] + abar[2] * ccs[2][k] + abar[3] * ccs[3][k]);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) ABt[r][c] = sg * (ccs[0][r] * G[0][c] + ccs[1][r] * G[1][c] + ccs[2][r] * G[2][c] + ccs[3][r] * G[3][c]);
        procrustes_rotation(ABt, Rk);
#pragma unroll
        for (int r = 0; r < 3; ++r) tk[r] = pc0[r] - (Rk[r][0] * c0[0] + Rk[r][1] * c0[1] + Rk[r][2] * c0[2]);
    };
    candidate(std::integral_constant<int, 0>{}, Rc[0], tc[0]);
    candidate(std::integral_constant<int, 1>{}, Rc[1], tc[1]);
    candidate(std::integral_constant<int, 2>{}, Rc[2], tc[2]);
    double err[3] = {0, 0, 0};
    for_points([&](int i) {
        double pw[3], uv[2];
        get(i, pw, uv);
#pragma unroll
        for (int na = 0; na < 3; ++na) {
            const double X = Rc[na][0][0] * pw[0] + Rc[na][0][1] * pw[1] + Rc[na][0][2] * pw[2] + tc[na][0];
            const double Y = Rc[na][1][0] * pw[0] + Rc[na][1][1] * pw[1] + Rc[na][1][2] * pw[2] + tc[na][1];
            const double Z = Rc[na][2][0] * pw[0] + Rc[na][2][1] * pw[1] + Rc[na][2][2] * pw[2] + tc[na][2];
            const double du = uv[0] - (cam.uc + cam.fu * X / Z), dvv = uv[1] - (cam.vc + cam.fv * Y / Z);
            err[na] += sqrt(du * du + dvv * dvv);
        }
    });
#pragma unroll
    for (int na = 0; na < 3; ++na) err[na] = red(err[na]);
    // smallest finite error, first on ties (N = 1, 2, 3 order)
    const bool f0 = isfinite(err[0]), f1 = isfinite(err[1]), f2 = isfinite(err[2]);
    if (!(f0 || f1 || f2)) return false;
    const bool pick1 = f1 && (!f0 || err[1] < err[0]);
    const double e01 = pick1 ? err[1] : err[0];
    const bool pick2 = f2 && (!(f0 || f1) || err[2] < e01);
    bool fin = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Rout[r][c] = pick2 ? Rc[2][r][c] : (pick1 ? Rc[1][r][c] : Rc[0][r][c]);
            fin &= isfinite(Rout[r][c]);
        }
        tout[r] = pick2 ? tc[2][r] : (pick1 ? tc[1][r] : tc[0][r]);
        fin &= isfinite(tout[r]);
    }
    return fin;
}

// ---- kernels --------------------------------------------------------------------------------------------------
// n_dev != nullptr: the number of correspondences is only known on the device (gather_matches_kernel)
__global__ __launch_bounds__(64) void hyp_kernel(const float* __restrict__ p3, const float* __restrict__ p2, int n_host,
                                                 const int* __restrict__ n_dev, double scale, Cam cam, unsigned long long seed,
                                                 int iterations, double* __restrict__ hyp) {
    const int h = blockIdx.x * 64 + threadIdx.x;
    if (h >= iterations) return;
    const int n = n_dev ? *n_dev : n_host;
    const double nan = __longlong_as_double(0x7FF8000000000000ll);
    if (n < MODEL_POINTS) {                                  // too few matches: no model (eval_utils.py:40-42)
        for (int k = 0; k < 12; ++k) hyp[(size_t)h * 12 + k] = nan;
        return;
    }
    int idx[MODEL_POINTS];
    sample_indices(seed, h, n, idx);
    double spw[MODEL_POINTS][3], suv[MODEL_POINTS][2];
    for (int k = 0; k < MODEL_POINTS; ++k) {
        for (int c = 0; c < 3; ++c) spw[k][c] = (double)p3[(size_t)idx[k] * 3 + c] * scale;
        suv[k][0] = (double)p2[(size_t)idx[k] * 2];
        suv[k][1] = (double)p2[(size_t)idx[k] * 2 + 1];
    }
    auto get = [&](int i, double (&pw)[3], double (&uv)[2]) {
        pw[0] = spw[i][0]; pw[1] = spw[i][1]; pw[2] = spw[i][2];
        uv[0] = suv[i][0]; uv[1] = suv[i][1];
    };
    double R[3][3], t[3];
    const bool ok = epnp_solve<MODEL_POINTS>(MODEL_POINTS, get, cam, R, t);
    double* o = hyp + (size_t)h * 12;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) o[r * 4 + c] = ok ? R[r][c] : nan;
        o[r * 4 + 3] = ok ? t[r] : nan;
    }
}

__device__ __forceinline__ bool is_inlier(const double* __restrict__ P, const float* __restrict__ p3, const float* __restrict__ p2,
                                          int i, double scale, const Cam& cam, double thr2) {
    const double x = (double)p3[(size_t)i * 3] * scale, y = (double)p3[(size_t)i * 3 + 1] * scale, z = (double)p3[(size_t)i * 3 + 2] * scale;
    const double X = P[0] * x + P[1] * y + P[2] * z + P[3];
    const double Y = P[4] * x + P[5] * y + P[6] * z + P[7];
    const double Z = P[8] * x + P[9] * y + P[10] * z + P[11];
    const double du = (double)p2[(size_t)i * 2] - (cam.uc + cam.fu * X / Z), dv = (double)p2[(size_t)i * 2 + 1] - (cam.vc + cam.fv * Y / Z);
    return du * du + dv * dv <= thr2;      // false for NaN poses / points at infinity
}

__global__ __launch_bounds__(256) void score_kernel(const float* __restrict__ p3, const float* __restrict__ p2, int n_host,
                                                    const int* __restrict__ n_dev, double scale, Cam cam, double thr2, int iterations,
                                                    const double* __restrict__ hyp, int* __restrict__ counts) {
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (h >= iterations) return;
    const int n = n_dev ? *n_dev : n_host;
    double P[12];
    for (int k = 0; k < 12; ++k) P[k] = hyp[(size_t)h * 12 + k];
    int cnt = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        const bool in = i < n && is_inlier(P, p3, p2, i, scale, cam, thr2);
        cnt += __popcll(__ballot(in));
    }
    if (lane == 0) counts[h] = cnt;
}

// src != nullptr: correspondence i came from query keypoint src[i]; the inlier mask is indexed by query keypoint
// (pre-zeroed by gather_matches_kernel)
__global__ __launch_bounds__(1024) void best_kernel(const float* __restrict__ p3, const float* __restrict__ p2, int n_host,
                                                    const int* __restrict__ n_dev, double scale, Cam cam, double thr2, int iterations,
                                                    const double* __restrict__ hyp, const int* __restrict__ counts,
                                                    const int* __restrict__ src, int32_t* __restrict__ mask,
                                                    int* __restrict__ inl_idx, int32_t* __restrict__ info) {
    __shared__ int sc[1024], si[1024], wsum[16];
    const int tid = threadIdx.x;
    const int n = n_dev ? *n_dev : n_host;
    int bc = -1, bi = 0x7FFFFFFF;
    for (int h = tid; h < iterations; h += 1024) {
        const int c = counts[h];
        if (c > bc) { bc = c; bi = h; }                    // ascending h per thread: first maximum kept
    }
    sc[tid] = bc; si[tid] = bi;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s && (sc[tid + s] > sc[tid] || (sc[tid + s] == sc[tid] && si[tid + s] < si[tid]))) {
            sc[tid] = sc[tid + s]; si[tid] = si[tid + s];
        }
        __syncthreads();
    }
    const int best = si[0], bcount = sc[0];
    const bool ok = bcount >= MODEL_POINTS;
    double P[12];
    for (int k = 0; k < 12; ++k) P[k] = ok ? hyp[(size_t)best * 12 + k] : 0.0;
    int run = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {                 // inlier mask + ordered index list
        const int i = i0 + tid;
        const int in = ok && i < n && is_inlier(P, p3, p2, i, scale, cam, thr2);
        if (i < n) mask[src ? src[i] : i] = in;
        const int lane = tid & 63, wave = tid >> 6;
        int inc = in;
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        __syncthreads();
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) base += wsum[w];
            tot += wsum[w];
        }
        if (in) inl_idx[run + base + inc - 1] = i;
        run += tot;
    }
    if (tid == 0) {
        info[0] = ok; info[1] = ok ? run : 0; info[2] = ok ? best : -1; info[3] = bcount < 0 ? 0 : bcount;
    }
}

// inference.py:148-152 on the device: valid = matches0 > -1; mkpts2d = kpts2d[valid]; mkpts3d = kpts3d[matches0[valid]] --
// ordered compaction by one workgroup; also zeroes the per-keypoint inlier mask
__global__ __launch_bounds__(1024) void gather_matches_kernel(const float* __restrict__ kpts2d, const float* __restrict__ kpts3d,
                                                              const long long* __restrict__ matches0, int n1, float* __restrict__ p2,
                                                              float* __restrict__ p3, int* __restrict__ src, int* __restrict__ count,
                                                              int32_t* __restrict__ mask) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int run = 0;
    for (int i0 = 0; i0 < n1; i0 += 1024) {
        const int i = i0 + tid;
        const long long m = i < n1 ? matches0[i] : -1;
        const int valid = m > -1;
        if (i < n1) mask[i] = 0;
        int inc = valid;
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(inc, d);
            if (lane >= d) inc += t;
        }
        __syncthreads();
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0, tot = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) base += wsum[w];
            tot += wsum[w];
        }
        if (valid) {
            const int o = run + base + inc - 1;
            p2[(size_t)o * 2] = kpts2d[(size_t)i * 2];
            p2[(size_t)o * 2 + 1] = kpts2d[(size_t)i * 2 + 1];
            for (int c = 0; c < 3; ++c) p3[(size_t)o * 3 + c] = kpts3d[(size_t)m * 3 + c];
            src[o] = i;
        }
        run += tot;
    }
    if (tid == 0) *count = run;
}

// EPnP over the listed correspondences (idx == nullptr: all n); info != nullptr: RANSAC refit (count from info[1])
__global__ __launch_bounds__(64) void refit_kernel(const float* __restrict__ p3, const float* __restrict__ p2, int n_all, double scale,
                                                   Cam cam, const int* __restrict__ idx, int32_t* __restrict__ info,
                                                   double* __restrict__ pose) {
    const int lane = threadIdx.x;
    bool ok = true;
    int n = n_all;
    if (info) { ok = info[0] != 0; n = info[1]; }
    double R[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, t[3] = {0, 0, 0};
    if (ok && n >= 4) {
        auto get = [&](int i, double (&pw)[3], double (&uv)[2]) {
            const int j = idx ? idx[i] : i;
            pw[0] = (double)p3[(size_t)j * 3] * scale; pw[1] = (double)p3[(size_t)j * 3 + 1] * scale; pw[2] = (double)p3[(size_t)j * 3 + 2] * scale;
            uv[0] = (double)p2[(size_t)j * 2]; uv[1] = (double)p2[(size_t)j * 2 + 1];
        };
        double Rs[3][3], ts[3];
        if (epnp_solve<0>(n, get, cam, Rs, ts)) {
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) R[r][c] = Rs[r][c];
                t[r] = ts[r] / scale;
            }
        } else ok = false;
    } else ok = false;
    if (lane == 0) {
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) pose[r * 4 + c] = R[r][c];
            pose[r * 4 + 3] = t[r];
        }
        if (info && !ok) { info[0] = 0; info[1] = 0; }
    }
}

struct Workspace {
    double* hyp;
    int *counts, *inl, *src, *count;
    float *p2, *p3;
    size_t bytes;
};
inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }
inline Workspace carve(void* base, int n, int iterations) {
    Workspace w;
    char* p = static_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t nb) { char* r = p ? p + off : nullptr; off += align_up(nb); return r; };
    w.hyp = (double*)take(sizeof(double) * 12 * (size_t)iterations);
    w.counts = (int*)take(sizeof(int) * (size_t)iterations);
    w.inl = (int*)take(sizeof(int) * (size_t)n);
    w.src = (int*)take(sizeof(int) * (size_t)n);
    w.count = (int*)take(sizeof(int));
    w.p2 = (float*)take(sizeof(float) * 2 * (size_t)n);
    w.p3 = (float*)take(sizeof(float) * 3 * (size_t)n);
    w.bytes = off;
    return w;
}

}  // namespace pnp

using namespace pnp;

namespace {
thread_local char g_err[512] = "";
int fail(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}
Cam cam_of(const double* K) { return Cam{K[0], K[4], K[2], K[5]}; }
}  // namespace

extern "C" {

int pnp_version(void) { return 1; }
const char* pnp_last_error(void) { return g_err; }

size_t pnp_workspace_bytes(int n, int iterations) {
    if (n < 1 || iterations < 1) { fail("n and iterations must be >= 1"); return 0; }
    return carve(nullptr, n, iterations).bytes;
}

int pnp_ransac_epnp(const float* pts_3d, const float* pts_2d, const double* K_host, double scale, int n, double reproj_error,
                    int iterations, uint64_t seed, double* pose, int32_t* inlier_mask, int32_t* info, void* workspace,
                    size_t workspace_bytes, pnp_stream_t stream) {
    if (!pts_3d || !pts_2d || !K_host || !pose || !inlier_mask || !info || !workspace) return fail("null argument");
    if (n < MODEL_POINTS) return fail("solvePnPRansac with EPNP needs at least %d correspondences (got %d)", MODEL_POINTS, n);
    if (iterations < 1 || iterations > (1 << 24)) return fail("iterations out of range");
    if (!(scale > 0.0) || !(reproj_error > 0.0)) return fail("scale and reproj_error must be positive");
    Workspace w = carve(workspace, n, iterations);
    if (workspace_bytes < w.bytes) return fail("workspace too small: %zu < %zu bytes", workspace_bytes, w.bytes);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const Cam cam = cam_of(K_host);
    const double thr2 = reproj_error * reproj_error;
    const int* nd = nullptr;
    hipLaunchKernelGGL(hyp_kernel, dim3((iterations + 63) / 64), dim3(64), 0, s, pts_3d, pts_2d, n, nd, scale, cam,
                       (unsigned long long)seed, iterations, w.hyp);
    hipLaunchKernelGGL(score_kernel, dim3((iterations + 3) / 4), dim3(256), 0, s, pts_3d, pts_2d, n, nd, scale, cam, thr2,
                       iterations, w.hyp, w.counts);
    hipLaunchKernelGGL(best_kernel, dim3(1), dim3(1024), 0, s, pts_3d, pts_2d, n, nd, scale, cam, thr2, iterations, w.hyp, w.counts,
                       nd, inlier_mask, w.inl, info);
    hipLaunchKernelGGL(refit_kernel, dim3(1), dim3(64), 0, s, pts_3d, pts_2d, n, scale, cam, w.inl, info, pose);
    return check_launch("pnp_ransac_epnp");
}

int pnp_ransac_epnp_matches(const float* kpts2d, const float* kpts3d, const int64_t* matches0, int n1, const double* K_host,
                            double scale, double reproj_error, int iterations, uint64_t seed, double* pose, int32_t* inlier_mask,
                            int32_t* info, void* workspace, size_t workspace_bytes, pnp_stream_t stream) {
    if (!kpts2d || !kpts3d || !matches0 || !K_host || !pose || !inlier_mask || !info || !workspace) return fail("null argument");
    if (n1 < 1) return fail("n1 must be >= 1");
    if (iterations < 1 || iterations > (1 << 24)) return fail("iterations out of range");
    if (!(scale > 0.0) || !(reproj_error > 0.0)) return fail("scale and reproj_error must be positive");
    Workspace w = carve(workspace, n1, iterations);
    if (workspace_bytes < w.bytes) return fail("workspace too small: %zu < %zu bytes", workspace_bytes, w.bytes);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const Cam cam = cam_of(K_host);
    const double thr2 = reproj_error * reproj_error;
    hipLaunchKernelGGL(gather_matches_kernel, dim3(1), dim3(1024), 0, s, kpts2d, kpts3d, reinterpret_cast<const long long*>(matches0),
                       n1, w.p2, w.p3, w.src, w.count, inlier_mask);
    hipLaunchKernelGGL(hyp_kernel, dim3((iterations + 63) / 64), dim3(64), 0, s, w.p3, w.p2, 0, w.count, scale, cam,
                       (unsigned long long)seed, iterations, w.hyp);
    hipLaunchKernelGGL(score_kernel, dim3((iterations + 3) / 4), dim3(256), 0, s, w.p3, w.p2, 0, w.count, scale, cam, thr2, iterations,
                       w.hyp, w.counts);
    hipLaunchKernelGGL(best_kernel, dim3(1), dim3(1024), 0, s, w.p3, w.p2, 0, w.count, scale, cam, thr2, iterations, w.hyp, w.counts,
                       w.src, inlier_mask, w.inl, info);
    hipLaunchKernelGGL(refit_kernel, dim3(1), dim3(64), 0, s, w.p3, w.p2, 0, scale, cam, w.inl, info, pose);
    return check_launch("pnp_ransac_epnp_matches");
}

int pnp_epnp(const float* pts_3d, const float* pts_2d, const double* K_host, double scale, int n, double* pose, void* workspace,
             size_t workspace_bytes, pnp_stream_t stream) {
    (void)workspace; (void)workspace_bytes;
    if (!pts_3d || !pts_2d || !K_host || !pose) return fail("null argument");
    if (n < 4) return fail("EPnP needs at least 4 correspondences (got %d)", n);
    if (!(scale > 0.0)) return fail("scale must be positive");
    hipLaunchKernelGGL(refit_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), pts_3d, pts_2d, n, scale,
                       cam_of(K_host), (const int*)nullptr, (int32_t*)nullptr, pose);
    return check_launch("pnp_epnp");
}

}  // extern "C"
