// HBM-streaming / reduction kernels of the GATsSPG forward: state load/store, the GATs leaf
// aggregation, the dual-softmax finalisation with row/column arg-max, the mutual-NN tail, the
// one-time weight packing and the (off-path) KeypointEncoder.
#include <math.h>
#include <stdlib.h>

#include "../../include/gatsspg.h"
#include "gatsspg_launch.h"

namespace gatsspg {

__device__ __forceinline__ float elu_f(float x) { return x > 0.f ? x : expm1f(x); }

// ------------------------------------------------------------------------------------------------------
// state load / store    (GATs_SuperGlue.py:192-193: the .float() descriptors become the GNN state)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void load_state_kernel(const float* __restrict__ dq, const float* __restrict__ d3,
                                                         float* __restrict__ Z, ColLayout L, int zero_missing) {
    // one workgroup per (frame, channel) row; float4 stores (np is a multiple of 128), float4 loads when the
    // source rows are 16-byte aligned (n1, n2 multiples of 4).  A null source leaves its side untouched
    // (zero_missing = 0) or zero-fills it (zero_missing = 1).
    const int ch = blockIdx.x, f = blockIdx.y;
    float* zr = Z + (size_t)ch * L.ld + (size_t)f * L.np;
    const float* q = dq ? dq + ((size_t)f * D + ch) * L.n1 : nullptr;
    const float* y = d3 ? d3 + ((size_t)f * D + ch) * L.n2 : nullptr;
    const bool vec = ((L.n1 | L.n2) & 3) == 0;
    for (int i = threadIdx.x * 4; i < L.np; i += 1024) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool side = i >= L.n1p;
        const int j = side ? i - L.n1p : i;
        const int n = side ? L.n2 : L.n1;
        const float* src = side ? y : q;
        if (!src) {
            if (!zero_missing) continue;
        } else if (vec && j + 3 < n) {
            v = *reinterpret_cast<const float4*>(src + j);
        } else {
            if (j < n) v.x = src[j];
            if (j + 1 < n) v.y = src[j + 1];
            if (j + 2 < n) v.z = src[j + 2];
            if (j + 3 < n) v.w = src[j + 3];
        }
        *reinterpret_cast<float4*>(zr + i) = v;
    }
}

__global__ __launch_bounds__(256) void store_state_kernel(const float* __restrict__ S, float* __restrict__ o2,
                                                          float* __restrict__ o3, ColLayout L) {
    const int ch = blockIdx.x, f = blockIdx.y;
    const float* zr = S + (size_t)ch * L.ld + (size_t)f * L.np;
    if (o2) {
        float* q = o2 + ((size_t)f * D + ch) * L.n1;
        for (int i = threadIdx.x; i < L.n1; i += 256) q[i] = zr[i];
    }
    if (o3) {
        float* y = o3 + ((size_t)f * D + ch) * L.n2;
        for (int j = threadIdx.x; j < L.n2; j += 256) y[j] = zr[L.n1p + j];
    }
}

void launch_load_state(const float* dq, const float* d3, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    GATSSPG_LAUNCH(hk, KID_LOAD_STATE, s, load_state_kernel, dim3(D, w.L.b), dim3(256), 0, s, dq, d3, w.Z, w.L, 1);
}
void launch_load_columns(const float* c2, const float* c3, float* dst, const Workspace& w, hipStream_t s) {
    hipLaunchKernelGGL(load_state_kernel, dim3(D, w.L.b), dim3(256), 0, s, c2, c3, dst, w.L, 0);
}
void launch_store_state(const float* src, float* out2d, float* out3d, const Workspace& w, hipStream_t s, ProfileHook*) {
    hipLaunchKernelGGL(store_state_kernel, dim3(D, w.L.b), dim3(256), 0, s, src, out2d, out3d, w.L);
}

// ------------------------------------------------------------------------------------------------------
// GATs layer (GraphAttentionLayer.forward, GATs.py:35-88) in its exact-algebra form:
//   logits need only  h . (W a[256:])  and  leaf . (W a[:256])   (u2, u1 folded at pack time), so the
//   layer is ONE streaming pass over the leaf descriptors: per 3D point softmax over (1+L) logits,
//   weighted sum of [h, leaves], elu.  No N*L x 256 x 256 GEMM.
// coefficient rules (c0 multiplies h, c_j the leaves):
//   include_self      : (c0, c_1..L) = softmax(LeakyReLU([2 s3, s3+s_1, .., s3+s_L]))    GATs.py:82-88,44
//                       additional (and no linear transform): c0 += 1                    GATs.py:61-62
//   not include_self  : c_j = softmax_L(LeakyReLU(s3+s_j)) / 2,  c0 = 1                  GATs.py:63-67
// raw_out (with_linear_transform): write the pre-activation aggregate; W^T and elu follow in a GEMM.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lrelu02(float x) { return x > 0.f ? x : 0.2f * x; }

// fast path, num_leaf == 8: 8 points (64 leaf columns) x 256 channels per workgroup, the whole
// [256 x 64] leaf tile lives in registers (16 float4 per thread), every HBM byte is read once.
__global__ __launch_bounds__(256) void gats_leaf8_kernel(const float* __restrict__ u1, const float* __restrict__ u2,
                                                         const float* __restrict__ leaves, const float* Z,
                                                         float* dst, ColLayout L, int flags, int raw_out) {
    // Z and dst alias when the layer updates the state in place (no __restrict__ on them)
    __shared__ float hs[D * 8];
    __shared__ float red3[4][8];
    __shared__ float redl[4][64];
    __shared__ float coef[8][9];
    const int f = blockIdx.y, n0 = blockIdx.x * 8;
    const int pv = min(8, L.n2 - n0);
    const size_t lrow = (size_t)L.n2 * 8;
    const float* Lf = leaves + (size_t)f * D * lrow + (size_t)n0 * 8;
    const size_t ycol = (size_t)f * L.np + L.n1p + n0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane >> 4, c4 = lane & 15;
    const int pt = c4 >> 1, lh = c4 & 1;
    const bool valid = pt < pv;

    float4 v[16];
    float u1r[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int ch = p * 16 + w * 4 + r;
        v[p] = valid ? *reinterpret_cast<const float4*>(Lf + (size_t)ch * lrow + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        u1r[p] = u1[ch];
    }
    // 3D-point descriptors h[ch][8 points] (state columns are padded, always in bounds)
    float hv[8];
    {
        const float4* zp = reinterpret_cast<const float4*>(Z + (size_t)tid * L.ld + ycol);
        const float4 a = zp[0], b = zp[1];
        hv[0] = a.x; hv[1] = a.y; hv[2] = a.z; hv[3] = a.w; hv[4] = b.x; hv[5] = b.y; hv[6] = b.z; hv[7] = b.w;
        const float u2v = u2[tid];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            hs[tid * 8 + i] = hv[i];
            float s = hv[i] * u2v;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) red3[w][i] = s;
        }
    }
    {
        float dl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            dl[0] += v[p].x * u1r[p]; dl[1] += v[p].y * u1r[p]; dl[2] += v[p].z * u1r[p]; dl[3] += v[p].w * u1r[p];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dl[j] += __shfl_xor(dl[j], 16);
            dl[j] += __shfl_xor(dl[j], 32);
        }
        if (lane < 16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) redl[w][c4 * 4 + j] = dl[j];
        }
    }
    __syncthreads();
    if (tid < 8) {
        const int include_self = flags & GATSSPG_FLAG_INCLUDE_SELF;
        const float s3 = (red3[0][tid] + red3[1][tid]) + (red3[2][tid] + red3[3][tid]);
        float e[9];
        e[0] = lrelu02(s3 + s3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = tid * 8 + j;
            e[1 + j] = lrelu02(s3 + ((redl[0][c] + redl[1][c]) + (redl[2][c] + redl[3][c])));
        }
        float m = include_self ? e[0] : e[1];
#pragma unroll
        for (int j = 1; j < 9; ++j) m = fmaxf(m, e[j]);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j > 0 || include_self) {
                e[j] = expf(e[j] - m);
                sum += e[j];
            }
        }
        if (include_self) {
            coef[tid][0] = e[0] / sum + ((flags & GATSSPG_FLAG_ADDITIONAL) && !raw_out ? 1.f : 0.f);
#pragma unroll
            for (int j = 1; j < 9; ++j) coef[tid][j] = e[j] / sum;
        } else {
            coef[tid][0] = 1.f;
#pragma unroll
            for (int j = 1; j < 9; ++j) coef[tid][j] = (e[j] / sum) / 2.f;
        }
    }
    __syncthreads();
    const float c0 = coef[pt][0];
    const float cj0 = coef[pt][1 + lh * 4 + 0], cj1 = coef[pt][1 + lh * 4 + 1];
    const float cj2 = coef[pt][1 + lh * 4 + 2], cj3 = coef[pt][1 + lh * 4 + 3];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int ch = p * 16 + w * 4 + r;
        float part = ((cj0 * v[p].x + cj1 * v[p].y) + (cj2 * v[p].z + cj3 * v[p].w));
        part += __shfl_xor(part, 1);
        if (lh == 0 && valid) {
            const float val = c0 * hs[ch * 8 + pt] + part;
            dst[(size_t)ch * L.ld + ycol + pt] = raw_out ? val : elu_f(val);
        }
    }
}

// num_leaf == 8, 4 points (32 leaf columns = one 128-byte line per channel row) per workgroup: 8 float4 per
// thread (~70 VGPRs -> 7 workgroups per CU, 224 KiB of loads in flight per CU) and 2x more workgroups
// than the 8-point variant (finer tail).  Consecutive tiles are mapped to the same XCD (workgroup id g
// runs on XCD g % 8) so the 16-byte output pieces of one 128-byte line meet in one L2.
__global__ __launch_bounds__(256, 7) void gats_leaf8x4_kernel(const float* __restrict__ u1, const float* __restrict__ u2,
                                                           const float* __restrict__ leaves, const float* Z, float* dst,
                                                           ColLayout L, int flags, int raw_out, int ntiles) {
    __shared__ float hs[D * 4];
    __shared__ float red3[4][4];
    __shared__ float redl[4][32];
    __shared__ float coef[4][9];
    const int f = blockIdx.y;
    // bijective XCD-contiguous remap (cdna guide T1, non-multiple-of-8 safe)
    int tile;
    {
        const int g = blockIdx.x, xcd = g & 7, q = ntiles >> 3, r = ntiles & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (g >> 3);
    }
    const int n0 = tile * 4;
    const int pv = min(4, L.n2 - n0);
    const size_t lrow = (size_t)L.n2 * 8;
    const float* Lf = leaves + (size_t)f * D * lrow + (size_t)n0 * 8;
    const size_t ycol = (size_t)f * L.np + L.n1p + n0;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane >> 3, c4 = lane & 7;
    const int pt = c4 >> 1, lh = c4 & 1;
    const bool valid = pt < pv;

    float4 v[8];
    float u1r[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int ch = p * 32 + w * 8 + r;
        v[p] = valid ? *reinterpret_cast<const float4*>(Lf + (size_t)ch * lrow + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        u1r[p] = u1[ch];
    }
    {
        const float4 a = *reinterpret_cast<const float4*>(Z + (size_t)tid * L.ld + ycol);
        const float hv[4] = {a.x, a.y, a.z, a.w};
        const float u2v = u2[tid];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hs[tid * 4 + i] = hv[i];
            float s = hv[i] * u2v;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) red3[w][i] = s;
        }
    }
    {
        float dl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            dl[0] += v[p].x * u1r[p]; dl[1] += v[p].y * u1r[p]; dl[2] += v[p].z * u1r[p]; dl[3] += v[p].w * u1r[p];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dl[j] += __shfl_xor(dl[j], 8);
            dl[j] += __shfl_xor(dl[j], 16);
            dl[j] += __shfl_xor(dl[j], 32);
        }
        if (lane < 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) redl[w][c4 * 4 + j] = dl[j];
        }
    }
    __syncthreads();
    if (tid < 4) {
        const int include_self = flags & GATSSPG_FLAG_INCLUDE_SELF;
        const float s3 = (red3[0][tid] + red3[1][tid]) + (red3[2][tid] + red3[3][tid]);
        float e[9];
        e[0] = lrelu02(s3 + s3);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = tid * 8 + j;
            e[1 + j] = lrelu02(s3 + ((redl[0][c] + redl[1][c]) + (redl[2][c] + redl[3][c])));
        }
        float m = include_self ? e[0] : e[1];
#pragma unroll
        for (int j = 1; j < 9; ++j) m = fmaxf(m, e[j]);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            if (j > 0 || include_self) {
                e[j] = expf(e[j] - m);
                sum += e[j];
            }
        }
        if (include_self) {
            coef[tid][0] = e[0] / sum + ((flags & GATSSPG_FLAG_ADDITIONAL) && !raw_out ? 1.f : 0.f);
#pragma unroll
            for (int j = 1; j < 9; ++j) coef[tid][j] = e[j] / sum;
        } else {
            coef[tid][0] = 1.f;
#pragma unroll
            for (int j = 1; j < 9; ++j) coef[tid][j] = (e[j] / sum) / 2.f;
        }
    }
    __syncthreads();
    const float c0 = coef[pt][0];
    const float cj0 = coef[pt][1 + lh * 4 + 0], cj1 = coef[pt][1 + lh * 4 + 1];
    const float cj2 = coef[pt][1 + lh * 4 + 2], cj3 = coef[pt][1 + lh * 4 + 3];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int ch = p * 32 + w * 8 + r;
        float part = ((cj0 * v[p].x + cj1 * v[p].y) + (cj2 * v[p].z + cj3 * v[p].w));
        part += __shfl_xor(part, 1);
        float val = c0 * hs[ch * 4 + pt] + part;
        if (!raw_out) val = elu_f(val);
        // gather the 4 points of this channel row into the lane with c4 == 0 -> one 16-byte store
        const float v1 = __shfl(val, (lane & ~7) | 2), v2 = __shfl(val, (lane & ~7) | 4), v3 = __shfl(val, (lane & ~7) | 6);
        if (c4 == 0) {
            float* o = dst + (size_t)ch * L.ld + ycol;
            if (pv == 4) {
                *reinterpret_cast<float4*>(o) = make_float4(val, v1, v2, v3);
            } else {
                o[0] = val;
                if (pv > 1) o[1] = v1;
                if (pv > 2) o[2] = v2;
            }
        }
    }
}

// generic path (any num_leaf <= 64): one thread per channel, 4 points per workgroup.  Off the
// benchmarked path; leaf reads are strided, but every configuration the reference accepts runs on
// the GPU (there is no CPU fallback).
constexpr int GATS_MAXL = 64;

__device__ __forceinline__ float block_sum_256(float x, float* scratch) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_xor(x, o);
    __syncthreads();  // protect scratch from the previous call's readers
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = x;
    __syncthreads();
    return (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
}

__global__ __launch_bounds__(256) void gats_generic_kernel(const float* __restrict__ u1, const float* __restrict__ u2,
                                                           const float* __restrict__ leaves, const float* Z,
                                                           float* dst, ColLayout L, int nl, int flags,
                                                           int raw_out) {
    __shared__ float scratch[4];
    __shared__ float sl[GATS_MAXL];
    __shared__ float coef[GATS_MAXL + 1];
    const int f = blockIdx.y, ch = threadIdx.x;
    const size_t lrow = (size_t)L.n2 * nl;
    const float* Lr = leaves + ((size_t)f * D + ch) * lrow;
    const float u1v = u1[ch], u2v = u2[ch];
    const int include_self = flags & GATSSPG_FLAG_INCLUDE_SELF;
    for (int pt = 0; pt < 4; ++pt) {
        const int n = blockIdx.x * 4 + pt;
        if (n >= L.n2) break;
        const size_t zc = (size_t)ch * L.ld + (size_t)f * L.np + L.n1p + n;
        const float hval = Z[zc];
        const float s3 = block_sum_256(hval * u2v, scratch);
        for (int j = 0; j < nl; ++j) {
            const float s = block_sum_256(Lr[(size_t)n * nl + j] * u1v, scratch);
            if (ch == 0) sl[j] = s;
        }
        __syncthreads();
        if (ch == 0) {
            const float e0 = lrelu02(s3 + s3);
            float m = include_self ? e0 : -INFINITY;
            for (int j = 0; j < nl; ++j) {
                sl[j] = lrelu02(s3 + sl[j]);
                m = fmaxf(m, sl[j]);
            }
            float sum = 0.f;
            float w0 = 0.f;
            if (include_self) {
                w0 = expf(e0 - m);
                sum = w0;
            }
            for (int j = 0; j < nl; ++j) {
                sl[j] = expf(sl[j] - m);
                sum += sl[j];
            }
            if (include_self) {
                coef[0] = w0 / sum + ((flags & GATSSPG_FLAG_ADDITIONAL) && !raw_out ? 1.f : 0.f);
                for (int j = 0; j < nl; ++j) coef[1 + j] = sl[j] / sum;
            } else {
                coef[0] = 1.f;
                for (int j = 0; j < nl; ++j) coef[1 + j] = (sl[j] / sum) / 2.f;
            }
        }
        __syncthreads();
        float val = 0.f;
        for (int j = 0; j < nl; ++j) val += coef[1 + j] * Lr[(size_t)n * nl + j];
        val += coef[0] * hval;
        dst[zc] = raw_out ? val : elu_f(val);
        __syncthreads();
    }
}

void launch_gats(const float* u1, const float* u2, const float* leaves, int num_leaf, int flags, float* dst,
                 const Workspace& w, hipStream_t s, ProfileHook* hk) {
    const int raw_out = (flags & GATSSPG_FLAG_WITH_LINEAR_TRANSFORM) ? 1 : 0;
    static const int variant = [] { const char* v = getenv("GATSSPG_GATS_TILE"); return v ? atoi(v) : 4; }();
    if (num_leaf == 8 && variant == 4) {
        const int nt = (w.L.n2 + 3) / 4;
        GATSSPG_LAUNCH(hk, KID_GATS, s, gats_leaf8x4_kernel, dim3(nt, w.L.b), dim3(256), 0, s, u1, u2, leaves, w.Z, dst, w.L,
                       flags, raw_out, nt);
    } else if (num_leaf == 8) {
        GATSSPG_LAUNCH(hk, KID_GATS, s, gats_leaf8_kernel, dim3((w.L.n2 + 7) / 8, w.L.b), dim3(256), 0, s, u1, u2, leaves, w.Z,
                       dst, w.L, flags, raw_out);
    } else {
        GATSSPG_LAUNCH(hk, KID_GATS, s, gats_generic_kernel, dim3((w.L.n2 + 3) / 4, w.L.b), dim3(256), 0, s, u1, u2, leaves,
                       w.Z, dst, w.L, num_leaf, flags, raw_out);
    }
}

// ------------------------------------------------------------------------------------------------------
// dual softmax finalisation + matching     (GATs_SuperGlue.py:218-237)
// ------------------------------------------------------------------------------------------------------
// row sums (over n2) and column sums (over n1) of E from the score kernel's per-tile partials.
// 1024 threads = 64 outputs x 16 partial-ranges; ranges are summed in order and combined in order.
__global__ __launch_bounds__(1024) void softmax_sums_kernel(const float* __restrict__ rowpart,
                                                            const float* __restrict__ colpart, float* __restrict__ rs,
                                                            float* __restrict__ cs, ColLayout L, int nct, int nrt) {
    __shared__ float red[16][64];
    const int f = blockIdx.y, el = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int nrb = L.n1p / 64;
    const bool rows = (int)blockIdx.x < nrb;
    const int idx = (rows ? blockIdx.x : blockIdx.x - nrb) * 64 + el;
    const int T = rows ? nct : nrt;
    const size_t stride = rows ? L.n1p : L.n2p;
    const float* src = (rows ? rowpart : colpart) + (size_t)f * T * stride + idx;
    const int per = (T + 15) / 16;
    const int tb = part * per, te = min(T, tb + per);
    float s = 0.f;
#pragma unroll 8
    for (int t = tb; t < te; ++t) s += src[(size_t)t * stride];
    red[part][el] = s;
    __syncthreads();
    if (part == 0) {
        float tot = red[0][el];
#pragma unroll
        for (int p = 1; p < 16; ++p) tot += red[p][el];
        (rows ? rs : cs)[(size_t)f * stride + idx] = tot;
    }
}

__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

// conf = softmax(S, dim=1) * softmax(S, dim=2) = (E / colsum) * (E / rowsum), in place over E;
// per 8-row strip the column (max, first arg-max row), per 1024-column chunk the row (max, first
// arg-max column).  torch.max on CPU breaks ties with the first index; so do we.
// VEC: n2 % 4 == 0 -> each thread owns 4 consecutive columns (16-byte accesses); otherwise columns
// tid + 256 k.
template <bool VEC>
__global__ __launch_bounds__(256) void conf_finalize_kernel(float* __restrict__ conf, const float* __restrict__ rs,
                                                            const float* __restrict__ cs, float* __restrict__ rmax_v,
                                                            int* __restrict__ rmax_i, float* __restrict__ cmax_v,
                                                            int* __restrict__ cmax_i, ColLayout L, int nch, int nst) {
    __shared__ float wv[CF_ROWS][4];
    __shared__ int wi[CF_ROWS][4];
    const int chk = blockIdx.x, st = blockIdx.y, f = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = st * CF_ROWS, j0 = chk * CF_COLS;
    float* cf = conf + (size_t)f * L.n1 * L.n2;
    int jc[4];
    float csj[4], cmv[4];
    int cmi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        jc[k] = VEC ? j0 + 4 * tid + k : j0 + tid + 256 * k;
        csj[k] = jc[k] < L.n2 ? cs[(size_t)f * L.n2p + jc[k]] : 1.f;
        cmv[k] = -INFINITY;
        cmi[k] = 0;
    }
    const int nrows = min(CF_ROWS, L.n1 - i0);
    // all CF_ROWS rows of the strip are fetched before any is processed: CF_ROWS x 16 B in flight per thread
    float e[CF_ROWS][4];
#pragma unroll
    for (int u = 0; u < CF_ROWS; ++u) {
        const size_t base = (size_t)(i0 + u) * L.n2;
        if (u < nrows) {
            if (VEC) {
                if (jc[0] < L.n2) {
                    const float4 x = *reinterpret_cast<const float4*>(cf + base + jc[0]);
                    e[u][0] = x.x; e[u][1] = x.y; e[u][2] = x.z; e[u][3] = x.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (jc[k] < L.n2) e[u][k] = cf[base + jc[k]];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < CF_ROWS; ++u) {
        const int r = u;
        if (r < nrows) {  // uniform over the block
            const int i = i0 + r;
            const size_t base = (size_t)i * L.n2;
            const float rsi = rs[(size_t)f * L.n1p + i];
            float rv = -INFINITY;
            int ri = 0x7fffffff;
            float c[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (jc[k] < L.n2) {
                    c[k] = (e[u][k] / csj[k]) * (e[u][k] / rsi);
                    if (c[k] > cmv[k]) { cmv[k] = c[k]; cmi[k] = i; }
                    if (c[k] > rv) { rv = c[k]; ri = jc[k]; }
                }
            }
            if (VEC) {
                if (jc[0] < L.n2) *reinterpret_cast<float4*>(cf + base + jc[0]) = make_float4(c[0], c[1], c[2], c[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (jc[k] < L.n2) cf[base + jc[k]] = c[k];
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const float ov = __shfl_xor(rv, o);
                const int oi = __shfl_xor(ri, o);
                argmax_combine(rv, ri, ov, oi);
            }
            if (lane == 0) { wv[r][wave] = rv; wi[r][wave] = ri; }
        }
    }
    __syncthreads();
    if (tid < nrows) {
        float v = wv[tid][0];
        int i = wi[tid][0];
        for (int q = 1; q < 4; ++q) argmax_combine(v, i, wv[tid][q], wi[tid][q]);
        rmax_v[((size_t)f * nch + chk) * L.n1p + i0 + tid] = v;
        rmax_i[((size_t)f * nch + chk) * L.n1p + i0 + tid] = i;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (jc[k] < L.n2) {
            cmax_v[((size_t)f * nst + st) * L.n2p + jc[k]] = cmv[k];
            cmax_i[((size_t)f * nst + st) * L.n2p + jc[k]] = cmi[k];
        }
    }
}

// max0 / indices0 (per query, over n2) and indices1 (per 3D point, over n1)   GATs_SuperGlue.py:220-221
// 64 outputs x 16 partial-ranges per block; partials are ordered by increasing index, ranges are
// scanned and combined in order with a strict '>' so the first arg-max wins.
__global__ __launch_bounds__(1024) void match_reduce_kernel(const float* __restrict__ rmax_v, const int* __restrict__ rmax_i,
                                                            const float* __restrict__ cmax_v, const int* __restrict__ cmax_i,
                                                            float* __restrict__ max0, int* __restrict__ idx0,
                                                            int* __restrict__ idx1, ColLayout L, int nch, int nst) {
    __shared__ float rv[16][64];
    __shared__ int ri[16][64];
    const int f = blockIdx.y, el = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int nrb = L.n1p / 64;
    const bool rows = (int)blockIdx.x < nrb;
    const int idx = (rows ? blockIdx.x : blockIdx.x - nrb) * 64 + el;
    const int T = rows ? nch : nst;
    const size_t stride = rows ? L.n1p : L.n2p;
    const float* pv = (rows ? rmax_v : cmax_v) + (size_t)f * T * stride + idx;
    const int* pi = (rows ? rmax_i : cmax_i) + (size_t)f * T * stride + idx;
    const bool live = idx < (rows ? L.n1 : L.n2);
    const int per = (T + 15) / 16;
    const int tb = part * per, te = min(T, tb + per);
    float v = -INFINITY;
    int a = 0;
    if (live) {
#pragma unroll 4
        for (int t = tb; t < te; ++t) {
            const float cv = pv[(size_t)t * stride];
            const int ci = pi[(size_t)t * stride];
            if (cv > v) { v = cv; a = ci; }
        }
    }
    rv[part][el] = v;
    ri[part][el] = a;
    __syncthreads();
    if (part == 0 && live) {
        v = rv[0][el];
        a = ri[0][el];
#pragma unroll
        for (int p = 1; p < 16; ++p)
            if (rv[p][el] > v) { v = rv[p][el]; a = ri[p][el]; }
        if (rows) {
            max0[(size_t)f * L.n1p + idx] = v;
            idx0[(size_t)f * L.n1p + idx] = a;
        } else {
            idx1[(size_t)f * L.n2p + idx] = a;
        }
    }
}

// mutual check, threshold, -1 fill                                       GATs_SuperGlue.py:222-237
__global__ __launch_bounds__(256) void match_tail_kernel(const float* __restrict__ max0, const int* __restrict__ idx0,
                                                         const int* __restrict__ idx1, float thr,
                                                         int64_t* __restrict__ matches0, int64_t* __restrict__ matches1,
                                                         float* __restrict__ ms0, float* __restrict__ ms1, ColLayout L) {
    const int f = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int* i0 = idx0 + (size_t)f * L.n1p;
    const int* i1 = idx1 + (size_t)f * L.n2p;
    const float* m0 = max0 + (size_t)f * L.n1p;
    if (idx < L.n1) {
        const int j = i0[idx];
        const bool mutual0 = i1[j] == idx;
        const float s0 = mutual0 ? m0[idx] : 0.f;
        const bool valid0 = mutual0 && s0 > thr;
        matches0[(size_t)f * L.n1 + idx] = valid0 ? (int64_t)j : (int64_t)-1;
        ms0[(size_t)f * L.n1 + idx] = s0;
    } else if (idx >= L.n1p && idx - L.n1p < L.n2) {
        const int j = idx - L.n1p;
        const int i = i1[j];
        const bool mutual1 = i0[i] == j;
        const bool mutual0_i = i1[i0[i]] == i;
        const float s0_i = mutual0_i ? m0[i] : 0.f;
        const float s1 = mutual1 ? s0_i : 0.f;
        const bool valid1 = mutual1 && (mutual0_i && s0_i > thr);
        matches1[(size_t)f * L.n2 + j] = valid1 ? (int64_t)i : (int64_t)-1;
        ms1[(size_t)f * L.n2 + j] = s1;
    }
}

void launch_dual_softmax_match(const Workspace& w, float* conf, float thr, int64_t* matches0, int64_t* matches1,
                               float* mscores0, float* mscores1, hipStream_t s, ProfileHook* hk) {
    const ColLayout& L = w.L;
    const dim3 g1((L.n1p + L.n2p + 255) / 256, L.b);
    const dim3 g64((L.n1p + L.n2p) / 64, L.b);
    GATSSPG_LAUNCH(hk, KID_SOFTMAX_SUMS, s, softmax_sums_kernel, g64, dim3(1024), 0, s, w.rowpart, w.colpart, w.rs, w.cs, L,
                   w.sc_nct, w.sc_nrt);
    if ((L.n2 & 3) == 0 && (reinterpret_cast<uintptr_t>(conf) & 15) == 0) {
        GATSSPG_LAUNCH(hk, KID_CONF_FINALIZE, s, conf_finalize_kernel<true>, dim3(w.cf_nch, w.cf_nst, L.b), dim3(256), 0, s,
                       conf, w.rs, w.cs, w.rmax_v, w.rmax_i, w.cmax_v, w.cmax_i, L, w.cf_nch, w.cf_nst);
    } else {
        GATSSPG_LAUNCH(hk, KID_CONF_FINALIZE, s, conf_finalize_kernel<false>, dim3(w.cf_nch, w.cf_nst, L.b), dim3(256), 0, s,
                       conf, w.rs, w.cs, w.rmax_v, w.rmax_i, w.cmax_v, w.cmax_i, L, w.cf_nch, w.cf_nst);
    }
    GATSSPG_LAUNCH(hk, KID_MATCH_REDUCE, s, match_reduce_kernel, g64, dim3(1024), 0, s, w.rmax_v, w.rmax_i, w.cmax_v,
                   w.cmax_i, w.max0, w.idx0, w.idx1, L, w.cf_nch, w.cf_nst);
    GATSSPG_LAUNCH(hk, KID_MATCH_TAIL, s, match_tail_kernel, g1, dim3(256), 0, s, w.max0, w.idx0, w.idx1, thr, matches0,
                   matches1, mscores0, mscores1, L);
}

// ------------------------------------------------------------------------------------------------------
// one-time weight packing
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_qkv_kernel(gatsspg_raw_weights raw, float* __restrict__ packed) {
    const int row = blockIdx.x, layer = blockIdx.y, c = threadIdx.x;
    float* dstW = packed + PW_ATTN + (size_t)layer * AttnW::SIZE + AttnW::WQKV;
    float* dstB = packed + PW_ATTN + (size_t)layer * AttnW::SIZE + AttnW::BQKV;
    int which, src;
    if (row < 256) {
        which = 0;
        src = (row % 64) * 4 + row / 64;  // head-major row h*64+d  <-  reference channel d*4+h  (:97)
    } else {
        const int rr = row - 256, h = rr / 128, x = rr % 128;
        which = x < 64 ? 1 : 2;
        src = (x % 64) * 4 + h;
    }
    dstW[(size_t)row * D + c] = raw.proj_w[layer][which][(size_t)src * D + c];
    if (c == 0) dstB[row] = raw.proj_b[layer][which][src];
}

// W0'[:, :256] = W0[:, :256];  W0'[:, 256 + (h*64+q)] = sum_o W0[:, 256+o] Wm[o][q*4+h];  b0' = b0 + W0b bm
__global__ __launch_bounds__(256) void pack_mlp0_kernel(gatsspg_raw_weights raw, float* __restrict__ packed) {
    const int row = blockIdx.x, layer = blockIdx.y, c = threadIdx.x;
    float* dstW = packed + PW_ATTN + (size_t)layer * AttnW::SIZE + AttnW::W0;
    float* dstB = packed + PW_ATTN + (size_t)layer * AttnW::SIZE + AttnW::B0;
    const float* W0 = raw.mlp0_w[layer] + (size_t)row * 512;
    const float* Wm = raw.merge_w[layer];
    dstW[(size_t)row * 512 + c] = W0[c];
    const int cref = (c % 64) * 4 + c / 64;
    double s = 0.0;
    for (int o = 0; o < 256; ++o) s += (double)W0[256 + o] * (double)Wm[(size_t)o * D + cref];
    dstW[(size_t)row * 512 + 256 + c] = (float)s;
    if (c == 0) {
        double sb = raw.mlp0_b[layer][row];
        for (int o = 0; o < 256; ++o) sb += (double)W0[256 + o] * (double)raw.merge_b[layer][o];
        dstB[row] = (float)sb;
    }
}

__global__ __launch_bounds__(256) void pack_rest_kernel(gatsspg_raw_weights raw, float* __restrict__ packed) {
    const int row = blockIdx.x, which = blockIdx.y, c = threadIdx.x;
    if (which < 8) {  // mlp.3
        float* dst = packed + PW_ATTN + (size_t)which * AttnW::SIZE;
        dst[AttnW::W3 + (size_t)row * 512 + c] = raw.mlp3_w[which][(size_t)row * 512 + c];
        dst[AttnW::W3 + (size_t)row * 512 + 256 + c] = raw.mlp3_w[which][(size_t)row * 512 + 256 + c];
        if (c == 0) dst[AttnW::B3 + row] = raw.mlp3_b[which][row];
    } else if (which < 12) {  // GATs layer: u1 = W a[:256], u2 = W a[256:]  (h @ W @ a == h . (W a))
        const int g = which - 8;
        float* dst = packed + PW_GATS + (size_t)g * GatsW::SIZE;
        const float* W = raw.gats_W[g];
        dst[GatsW::W + (size_t)row * D + c] = W[(size_t)row * D + c];
        if (c < 2) {
            double s = 0.0;
            for (int o = 0; o < 256; ++o) s += (double)W[(size_t)row * D + o] * (double)raw.gats_a[g][c * 256 + o];
            dst[(c == 0 ? GatsW::U1 : GatsW::U2) + row] = (float)s;
        }
    } else {  // final_proj
        packed[PW_FINAL_W + (size_t)row * D + c] = raw.final_w[(size_t)row * D + c];
        if (c == 0) packed[PW_FINAL_B + row] = raw.final_b[row];
    }
}

void launch_pack_weights(const void* raw_struct_host, float* packed, hipStream_t s) {
    const gatsspg_raw_weights raw = *static_cast<const gatsspg_raw_weights*>(raw_struct_host);
    hipLaunchKernelGGL(pack_qkv_kernel, dim3(768, 8), dim3(256), 0, s, raw, packed);
    hipLaunchKernelGGL(pack_mlp0_kernel, dim3(512, 8), dim3(256), 0, s, raw, packed);
    hipLaunchKernelGGL(pack_rest_kernel, dim3(256, 13), dim3(256), 0, s, raw, packed);
}

// ------------------------------------------------------------------------------------------------------
// KeypointEncoder (GATs_SuperGlue.py:131-140).  Dead code in the reference forward (SURVEY.md section 0
// item 1); provided as a standalone op.  Tiny channel counts (3/4 -> 32 -> 64 -> 128 -> 256): plain
// FMA kernels, one thread per output element; InstanceNorm + ReLU in place per (sample, channel).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kenc_input_kernel(const float* __restrict__ kpts, const float* __restrict__ scores,
                                                         float* __restrict__ x0, int n, int kd) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        for (int c = 0; c < kd; ++c) x0[((size_t)b * (kd + 1) + c) * n + i] = kpts[((size_t)b * n + i) * kd + c];
        x0[((size_t)b * (kd + 1) + kd) * n + i] = scores[(size_t)b * n + i];
    }
}

__global__ __launch_bounds__(256) void kenc_conv_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ x, float* __restrict__ y, int cin,
                                                        int cout, int n) {
    const int o = blockIdx.y, b = blockIdx.z;
    const float* wr = w + (size_t)o * cin;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float s = 0.f;
        for (int c = 0; c < cin; ++c) s = fmaf(wr[c], x[((size_t)b * cin + c) * n + i], s);
        y[((size_t)b * cout + o) * n + i] = s + bias[o];
    }
}

__global__ __launch_bounds__(256) void kenc_in_relu_kernel(float* __restrict__ y, int n) {
    __shared__ float scratch[4];
    float* row = y + (size_t)blockIdx.x * n;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += row[i];
    const float mean = block_sum_256(s, scratch) / n;
    float s2 = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = row[i] - mean;
        s2 += d * d;
    }
    const float var = block_sum_256(s2, scratch) / n;
    const float rstd = 1.f / sqrtf(var + 1e-5f);
    for (int i = threadIdx.x; i < n; i += 256) row[i] = fmaxf((row[i] - mean) * rstd, 0.f);
}

size_t kenc_scratch_bytes(int b, int n) { return sizeof(float) * (size_t)b * n * (4 + 128 + 128); }

void launch_kenc(const float* const* w, const float* const* bias, int inp_dim, const float* kpts, const float* scores, int b,
                 int n, float* out, void* scratch, hipStream_t s) {
    float* x0 = static_cast<float*>(scratch);
    float* bufA = x0 + (size_t)b * n * 4;
    float* bufB = bufA + (size_t)b * n * 128;
    const int gx = (n + 255) / 256;
    hipLaunchKernelGGL(kenc_input_kernel, dim3(gx, b), dim3(256), 0, s, kpts, scores, x0, n, inp_dim - 1);
    const int chans[5] = {inp_dim, 32, 64, 128, 256};
    const float* cur = x0;
    float* bufs[2] = {bufA, bufB};
    for (int l = 0; l < 4; ++l) {
        float* y = l == 3 ? out : bufs[l & 1];
        hipLaunchKernelGGL(kenc_conv_kernel, dim3(gx, chans[l + 1], b), dim3(256), 0, s, w[l], bias[l], cur, y, chans[l],
                           chans[l + 1], n);
        if (l < 3) hipLaunchKernelGGL(kenc_in_relu_kernel, dim3(b * chans[l + 1]), dim3(256), 0, s, y, n);
        cur = y;
    }
}

}  // namespace gatsspg
