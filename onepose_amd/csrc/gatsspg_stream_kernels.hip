// HBM-streaming / reduction kernels of the GATsSPG forward: state load/store, the GATs leaf
// aggregation, the dual-softmax finalisation with row/column arg-max, the mutual-NN tail, the
// one-time weight packing and the (off-path) KeypointEncoder.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/gatsspg.h"
#include "gatsspg_launch.h"

namespace gatsspg {

int tuning_knob(const char* name, int dflt) {
#ifdef GATSSPG_TUNING
    char key[96];
    snprintf(key, sizeof(key), "GATSSPG_%s", name);
    const char* v = getenv(key);
    return v ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

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

// num_leaf == 8, 4 points (32 leaf columns = one 128-byte line per channel row) per workgroup: 8 float4 per
// thread (~70 VGPRs -> 7 workgroups per CU, 224 KiB of loads in flight per CU) and 2x more workgroups
// than the 8-point variant (finer tail).  Consecutive tiles are mapped to the same XCD (workgroup id g
// runs on XCD g % 8) so the 16-byte output pieces of one 128-byte line meet in one L2.
// Fused state load (first layer of a forward): h3 != nullptr -> the 3D-point descriptors are read straight from the
// caller's compact [b,256,n2] tensor instead of the state, and GATS_COPY_BLOCKS spare workgroups per frame copy the
// query descriptors dq [b,256,n1] into the 2D side of the state (pads zeroed) and zero the 3D-side pad columns --
// what load_state_kernel would have done in a launch of its own.
constexpr int GATS_COPY_BLOCKS = 64;   // 4 channel rows each

// leaf logits of one lane: its 4 leaves x its 8 channel rows, then over the 8 row lanes (lane bits 3..5) with two halving
// exchanges and one butterfly step: afterwards a lane holds leaf ((lane >> 3) & 1) * 2 + ((lane >> 4) & 1) of its 16-byte
// piece, summed over this wave's 64 channel rows.  ONE function for the layer kernel and for the per-object logit cache
// (gats_leaf_logits_kernel): the cached values are the bits the layer kernel would have computed.
__device__ __forceinline__ float leaf_logit_wave_partial(const float4 (&v)[8], const float (&u1r)[8], int lane) {
    float dl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        dl[0] += v[p].x * u1r[p]; dl[1] += v[p].y * u1r[p]; dl[2] += v[p].z * u1r[p]; dl[3] += v[p].w * u1r[p];
    }
    const bool b3 = lane & 8, b4 = lane & 16;
    float k0 = b3 ? dl[2] : dl[0], k1 = b3 ? dl[3] : dl[1];
    k0 += __shfl_xor(b3 ? dl[0] : dl[2], 8);
    k1 += __shfl_xor(b3 ? dl[1] : dl[3], 8);
    float s = b4 ? k1 : k0;
    s += __shfl_xor(b4 ? k0 : k1, 16);
    s += __shfl_xor(s, 32);
    return s;
}
__device__ __forceinline__ int leaf_logit_slot(int lane) { return (lane & 7) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 4) & 1); }

// CACHED_LOGITS: the leaf logits come from the per-object cache `cl` ([tiles][32] floats of this frame and layer) instead of
// being recomputed from the leaves (amortised mode)
template <bool FUSED_LOAD, bool CACHED_LOGITS>
__global__ __launch_bounds__(256, 7) void gats_leaf8x4_kernel(const float* __restrict__ u1, const float* __restrict__ u2,
                                                           const float* __restrict__ leaves, const float* Z, float* dst,
                                                           ColLayout L, int flags, int raw_out, int ntiles,
                                                           const float* __restrict__ h3, const float* __restrict__ dq,
                                                           const float* __restrict__ cl) {
    __shared__ __attribute__((aligned(16))) float hs[D * 4];
    __shared__ __attribute__((aligned(16))) float pre[D * 4];
    __shared__ float red3[4][4];
    __shared__ float redl[4][32];
    __shared__ float coef[4][9];
    const int f = blockIdx.y;
    if (FUSED_LOAD && (int)blockIdx.x >= ntiles) {   // state-load role
        const int ch = ((int)blockIdx.x - ntiles) * 4 + (threadIdx.x >> 6), ln = threadIdx.x & 63;
        float* zr = dst + (size_t)ch * L.ld + (size_t)f * L.np;
        const float* q = dq + ((size_t)f * D + ch) * L.n1;
        const bool vec = (L.n1 & 3) == 0;
        for (int i = ln * 4; i < L.n1p; i += 256) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vec && i + 3 < L.n1) {
                v = *reinterpret_cast<const float4*>(q + i);
            } else {
                if (i < L.n1) v.x = q[i];
                if (i + 1 < L.n1) v.y = q[i + 1];
                if (i + 2 < L.n1) v.z = q[i + 2];
                if (i + 3 < L.n1) v.w = q[i + 3];
            }
            *reinterpret_cast<float4*>(zr + i) = v;
        }
        for (int i = L.n2 + ln; i < L.n2p; i += 64) zr[L.n1p + i] = 0.f;
        return;
    }
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

    // the 3D-point descriptors are requested FIRST: loads retire in order, so the h-only work below (s3, the LDS copy of h)
    // waits for one load instead of for the whole leaf tile and runs while the leaves stream in
    float4 a;
    if (FUSED_LOAD) {
        const float* hp = h3 + ((size_t)f * D + tid) * L.n2 + n0;
        if (pv == 4 && (L.n2 & 3) == 0) {
            a = *reinterpret_cast<const float4*>(hp);
        } else {
            a.x = hp[0];
            a.y = pv > 1 ? hp[1] : 0.f;
            a.z = pv > 2 ? hp[2] : 0.f;
            a.w = pv > 3 ? hp[3] : 0.f;
        }
    } else {
        a = *reinterpret_cast<const float4*>(Z + (size_t)tid * L.ld + ycol);
    }
    const float u2v = u2[tid];
    float clv = 0.f;
    if (CACHED_LOGITS && tid < 32) clv = cl[((size_t)f * ntiles + tile) * 32 + tid];
    // UNCONDITIONAL loads: lanes of points beyond n2 (last tile only) re-read point 0 of the tile -- finite values whose
    // results are never stored.  (A `valid ? load : 0` form compiles to exec-masked branches with an s_waitcnt vmcnt(0)
    // in the middle of the sequence: the 8 loads of a lane were in flight 3 + 5 instead of 8 at a time.)
    const int c4l = valid ? c4 : (c4 & 1);
    float4 v[8];
    float u1r[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int ch = p * 32 + w * 8 + r;
        v[p] = *reinterpret_cast<const float4*>(Lf + (size_t)ch * lrow + 4 * c4l);
        u1r[p] = CACHED_LOGITS ? 0.f : u1[ch];
    }
#ifdef GATSSPG_PROFILING_BUILD
    if (raw_out == 2) {   // read-only probe (profiling builds, wrong results): the leaf stream alone, one 4-byte store per thread
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) acc += (v[p].x + v[p].y) + (v[p].z + v[p].w);
        if (acc == 1.2345e30f) dst[tid] = acc;
        return;
    }
#endif
    {
        // s3[i] = sum over the 256 channels of h[ch][i] * u2[ch]: 4 values per lane are reduced over the 64 lanes with 7
        // shuffles (two halving exchanges leave one value per lane -- point (lane & 1) * 2 + ((lane >> 1) & 1) -- then a
        // 4-step butterfly), not 4 x 6
        *reinterpret_cast<float4*>(hs + tid * 4) = a;
        const float p0 = a.x * u2v, p1 = a.y * u2v, p2 = a.z * u2v, p3 = a.w * u2v;
        const bool b0 = lane & 1, b1 = lane & 2;
        float k0 = b0 ? p2 : p0, k1 = b0 ? p3 : p1;
        k0 += __shfl_xor(b0 ? p0 : p2, 1);
        k1 += __shfl_xor(b0 ? p1 : p3, 1);
        float s = b1 ? k1 : k0;
        s += __shfl_xor(b1 ? k0 : k1, 2);
#pragma unroll
        for (int o = 4; o <= 32; o <<= 1) s += __shfl_xor(s, o);
        if (lane < 4) red3[w][(lane & 1) * 2 + (lane >> 1)] = s;
    }
    if (CACHED_LOGITS) {
        if (tid < 32) redl[0][tid] = clv;
    } else {
        const float s = leaf_logit_wave_partial(v, u1r, lane);
        if (lane < 32) redl[w][leaf_logit_slot(lane)] = s;
    }
    __syncthreads();
    if (tid < 64) {
        // softmax over the 1 + 8 logits of each of the 4 points on 4 x 16 lanes of wave 0: lane = point * 16 + j, j = 0 the point itself,
        // 1..8 its leaves (round 4 ran it on 4 threads with 9 logits each: ~250 serial instructions on every workgroup's critical path and a
        // 12-byte scratch spill under the kernel's 72-register budget).  Same values: max is order-free, and every lane adds the nine
        // exponentials in the order 0, 1, ..., 8 that the four-thread form used (excluded entries contribute an exact 0).
        const int include_self = flags & GATSSPG_FLAG_INCLUDE_SELF;
        const int p4 = lane >> 4, j = lane & 15;
        const bool in = j < 9 && (j > 0 || include_self);
        const float s3 = (red3[0][p4] + red3[1][p4]) + (red3[2][p4] + red3[3][p4]);
        const int c = p4 * 8 + min(max(j - 1, 0), 7);
        const float ll = CACHED_LOGITS ? redl[0][c] : (redl[0][c] + redl[1][c]) + (redl[2][c] + redl[3][c]);
        const float e = lrelu02(s3 + (j == 0 ? s3 : ll));
        float m = in ? e : -3.0e38f;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float ex = in ? expf(e - m) : 0.f;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) sum += __shfl(ex, (lane & 48) + k);
        if (j < 9) {
            float cf;
            if (include_self) cf = ex / sum + (j == 0 && (flags & GATSSPG_FLAG_ADDITIONAL) && !raw_out ? 1.f : 0.f);
            else cf = j == 0 ? 1.f : (ex / sum) / 2.f;
            coef[p4][j] = cf;
        }
    }
    __syncthreads();
    // weighted sum of the leaves: the two 16-byte halves of a point meet in the even lane, which parks the row's value in
    // LDS; the channel's thread then adds the h term, applies elu to its 4 points and stores 16 bytes (4 elu per thread
    // instead of 8, no gather shuffles)
    const float cj0 = coef[pt][1 + lh * 4 + 0], cj1 = coef[pt][1 + lh * 4 + 1];
    const float cj2 = coef[pt][1 + lh * 4 + 2], cj3 = coef[pt][1 + lh * 4 + 3];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int ch = p * 32 + w * 8 + r;
        float part = ((cj0 * v[p].x + cj1 * v[p].y) + (cj2 * v[p].z + cj3 * v[p].w));
        part += __shfl_xor(part, 1);
        if (lh == 0) pre[ch * 4 + pt] = part;
    }
    __syncthreads();
    {
        const float4 pr = *reinterpret_cast<const float4*>(pre + tid * 4);
        const float4 h4 = *reinterpret_cast<const float4*>(hs + tid * 4);
        float o0 = coef[0][0] * h4.x + pr.x, o1 = coef[1][0] * h4.y + pr.y, o2 = coef[2][0] * h4.z + pr.z, o3 = coef[3][0] * h4.w + pr.w;
        if (!raw_out) {
            o0 = elu_f(o0); o1 = elu_f(o1); o2 = elu_f(o2); o3 = elu_f(o3);
        }
        float* o = dst + (size_t)tid * L.ld + ycol;
        if (pv == 4) {
            *reinterpret_cast<float4*>(o) = make_float4(o0, o1, o2, o3);
        } else {
            o[0] = o0;
            if (pv > 1) o[1] = o1;
            if (pv > 2) o[2] = o2;
        }
    }
}

// Per-object leaf-logit cache: the logits leaf . u1 of GATs layers `first_layer` .. `first_layer + nlayers - 1` depend only
// on the database (the leaves never change, GATs_SuperGlue.py:53-54), so gatsspg_prepare_database computes them once with
// the layer kernel's own thread mapping and reduction order.  cl: [nlayers][b][tiles][32] floats.
__global__ __launch_bounds__(256, 7) void gats_leaf_logits_kernel(const float* __restrict__ u1_first, int u1_stride, int nlayers,
                                                                const float* __restrict__ leaves, float* __restrict__ cl,
                                                                ColLayout L, int ntiles) {
    __shared__ float redl[4][32];
    const int f = blockIdx.y, tile = blockIdx.x;
    const int n0 = tile * 4, pv = min(4, L.n2 - n0);
    const size_t lrow = (size_t)L.n2 * 8;
    const float* Lf = leaves + (size_t)f * D * lrow + (size_t)n0 * 8;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane >> 3, c4 = lane & 7;
    const int c4l = (c4 >> 1) < pv ? c4 : (c4 & 1);
    float4 v[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) v[p] = *reinterpret_cast<const float4*>(Lf + (size_t)(p * 32 + w * 8 + r) * lrow + 4 * c4l);
    for (int t = 0; t < nlayers; ++t) {
        float u1r[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) u1r[p] = u1_first[(size_t)t * u1_stride + p * 32 + w * 8 + r];
        const float s = leaf_logit_wave_partial(v, u1r, lane);
        if (lane < 32) redl[w][leaf_logit_slot(lane)] = s;
        __syncthreads();
        if (tid < 32)
            cl[(((size_t)t * L.b + f) * ntiles + tile) * 32 + tid] = (redl[0][tid] + redl[1][tid]) + (redl[2][tid] + redl[3][tid]);
        __syncthreads();
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

bool gats_fuses_state_load(int num_leaf, int flags, const Workspace& w) {
    return num_leaf == 8 && !(flags & GATSSPG_FLAG_WITH_LINEAR_TRANSFORM) && w.L.tw_first == 0 && w.L.tw_count == w.L.np / 64;
}

bool gats_caches_leaf_logits(int num_leaf, int flags) {
    return num_leaf == 8 && !(flags & GATSSPG_FLAG_WITH_LINEAR_TRANSFORM);
}
size_t gats_leaf_logit_floats(int b, int n2) { return (size_t)b * ((n2 + 3) / 4) * 32; }   // per layer

void launch_gats_leaf_logits(const float* u1_first, int u1_stride, int nlayers, const float* leaves, float* cl,
                             const Workspace& w, hipStream_t s) {
    const int nt = (w.L.n2 + 3) / 4;
    hipLaunchKernelGGL(gats_leaf_logits_kernel, dim3(nt, w.L.b), dim3(256), 0, s, u1_first, u1_stride, nlayers, leaves, cl, w.L, nt);
}

void launch_gats(const float* u1, const float* u2, const float* leaves, int num_leaf, int flags, float* dst,
                 const Workspace& w, hipStream_t s, ProfileHook* hk, const float* h3, const float* dq, const float* cl) {
    int raw_out = (flags & GATSSPG_FLAG_WITH_LINEAR_TRANSFORM) ? 1 : 0;
#ifdef GATSSPG_PROFILING_BUILD
    if (tuning_knob("GATS_PROBE", 0) && !h3) raw_out = 2;   // time the leaf stream alone
#endif
    if (num_leaf == 8) {
        const int nt = (w.L.n2 + 3) / 4;
        const int extra = h3 ? GATS_COPY_BLOCKS : 0;
        // (capping the residency -- fewer, staggered rounds of workgroups -- was measured: no gain at 4-5 per CU, worse below)
        if (h3)
            GATSSPG_LAUNCH(hk, KID_GATS, s, (gats_leaf8x4_kernel<true, false>), dim3(nt + extra, w.L.b), dim3(256), 0, s, u1, u2,
                           leaves, w.Z, dst, w.L, flags, raw_out, nt, h3, dq, cl);
        else if (cl)
            GATSSPG_LAUNCH(hk, KID_GATS, s, (gats_leaf8x4_kernel<false, true>), dim3(nt, w.L.b), dim3(256), 0, s, u1, u2, leaves,
                           w.Z, dst, w.L, flags, raw_out, nt, h3, dq, cl);
        else
            GATSSPG_LAUNCH(hk, KID_GATS, s, (gats_leaf8x4_kernel<false, false>), dim3(nt, w.L.b), dim3(256), 0, s, u1, u2, leaves,
                           w.Z, dst, w.L, flags, raw_out, nt, h3, dq, cl);
    } else {
        GATSSPG_LAUNCH(hk, KID_GATS, s, gats_generic_kernel, dim3((w.L.n2 + 3) / 4, w.L.b), dim3(256), 0, s, u1, u2, leaves,
                       w.Z, dst, w.L, num_leaf, flags, raw_out);
    }
}

// ------------------------------------------------------------------------------------------------------
// dual softmax finalisation + matching     (GATs_SuperGlue.py:218-237)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

// conf = softmax(S, dim=1) * softmax(S, dim=2)  (GATs_SuperGlue.py:218), in place over the conf buffer, for a strip of
// CF_ROWS rows x a chunk of CF_COLS columns per workgroup; per strip the column (max, first arg-max row), per chunk the
// row (max, first arg-max column).  torch.max on CPU breaks ties with the first index; so do we.
//   SHIFTED = false: the buffer holds E = exp(S) (score_exp_kernel); conf = (E / colsum) * (E / rowsum).  The two
//     normalisers are summed here, in the prologue, from the score kernel's per-tile partials (rows: this strip's
//     CF_ROWS sums of nct partials; columns: this chunk's CF_COLS sums of nrt partials) -- no reduction launch.
//   SHIFTED = true (1 / scale_factor > 80): the buffer holds S; conf = exp(S - colmax)/colsum * exp(S - rowmax)/rowsum
//     with the maxima / sums precomputed by softmax_rowstat / softmax_colstat (the max-subtracting softmax, total range).
// Each wave owns 4 whole rows of the strip (CF_COLS / 64 columns per lane and row; the 4 x CF_NQ 16-byte loads of a lane
// are issued first and fly while the normalisers are summed): the row arg-max needs one butterfly per row and wave, the
// column arg-max is combined across the 4 waves through LDS.
// VEC: n2 % 4 == 0 and conf 16-byte aligned -> a lane owns CF_NQ x 4 consecutive columns; otherwise columns lane + 64 k.
constexpr int CF_NQ = CF_COLS / 256;
// SMALL: at most 8 row partials per range (nct <= 128, i.e. n2 <= 8192) and one column partial per range (nrt <= 16, n1 <=
// 2048): the prologue is then straight-line code whose loads are all issued up front.  Larger problems take the looped
// prologue (hipcc drains every outstanding load -- the rows too -- in front of a loop that contains loads).
template <bool VEC, bool SHIFTED, bool SMALL>
__global__ __launch_bounds__(256) void conf_finalize_kernel(float* __restrict__ conf, const float* __restrict__ rowpart,
                                                            const float* __restrict__ colpart, const float* __restrict__ rs_g,
                                                            const float* __restrict__ cs_g, const float* __restrict__ rshift,
                                                            const float* __restrict__ cshift, float* __restrict__ rmax_v,
                                                            int* __restrict__ rmax_i, float* __restrict__ cmax_v,
                                                            int* __restrict__ cmax_i, ColLayout L, int nct, int nrt, int nch,
                                                            int nst) {
    __shared__ __attribute__((aligned(16))) float cs_s[CF_COLS];
    __shared__ __attribute__((aligned(16))) float csh_s[SHIFTED ? CF_COLS : 4];
    __shared__ float red[16][CF_ROWS];
    __shared__ float rs_s[CF_ROWS], rsh_s[CF_ROWS];
    __shared__ __attribute__((aligned(16))) float cmv_s[4][CF_COLS];
    __shared__ __attribute__((aligned(16))) int cmi_s[4][CF_COLS];
    const int chk = blockIdx.x, st = blockIdx.y, f = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = st * CF_ROWS, j0 = chk * CF_COLS;
    const int nrows = min(CF_ROWS, L.n1 - i0);
    float* cf = conf + (size_t)f * L.n1 * L.n2;

    // Load order = wait order (loads retire in order): first the partials of the normalisers into registers, then this wave's
    // 4 rows (every 16-byte load of a lane), then the sums -- they wait for the partials only, the rows keep streaming.
    auto lcol = [&](int q, int k) { return VEC ? 4 * lane + 256 * q + k : lane + 64 * (4 * q + k); };   // < CF_COLS
    const int pr_r = tid & (CF_ROWS - 1), pr_g = tid / CF_ROWS;   // row partials: 16 rows x 16 ranges of the nct partials
    const int pr_per = (nct + 15) / 16;
    const int pr_tb = pr_g * pr_per, pr_te = min(nct, pr_tb + pr_per);
    const float* pr_src = rowpart + (size_t)f * nct * L.n1p + i0 + pr_r;
    const int pc_jl = 4 * tid, pc_j = j0 + pc_jl;                  // column partials: 4 consecutive columns per thread
    const bool pc_on = pc_jl < CF_COLS && pc_j < L.n2p;            // n2p is a multiple of 128: all four in bounds
    const float* pc_src = colpart + (size_t)f * nrt * L.n2p + pc_j;
    const int pc_per = (nrt + 15) / 16;
    float xr[8];
    float4 xc[16];
    if constexpr (!SHIFTED && SMALL) {
        // every load below is UNCONDITIONAL on a clamped (always valid) address; out-of-range values are discarded afterwards.
        // (Guarded loads compile to exec-masked branches, and the waits hipcc places around those are vmcnt(0): the rows then
        // cannot stream under the sums.)
        const float* pr_row = rowpart + (size_t)f * nct * L.n1p + i0 + min(pr_r, nrows - 1);
#pragma unroll
        for (int u = 0; u < 8; ++u) xr[u] = pr_row[(size_t)min(pr_tb + u, nct - 1) * L.n1p];
        const float* pc_col = colpart + (size_t)f * nrt * L.n2p + min(pc_j, L.n2p - 4);
#pragma unroll
        for (int t = 0; t < 16; ++t) xc[t] = *reinterpret_cast<const float4*>(pc_col + (size_t)min(t, nrt - 1) * L.n2p);
    }
    float e[4][CF_NQ][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const size_t base = (size_t)(i0 + min(4 * wave + u, nrows - 1)) * L.n2;
#pragma unroll
        for (int q = 0; q < CF_NQ; ++q) {
            if (VEC) {
                const float4 x = *reinterpret_cast<const float4*>(cf + base + min(j0 + lcol(q, 0), L.n2 - 4));
                e[u][q][0] = x.x; e[u][q][1] = x.y; e[u][q][2] = x.z; e[u][q][3] = x.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) e[u][q][k] = cf[base + min(j0 + lcol(q, k), L.n2 - 1)];
            }
        }
    }
    // ---- normalisers of this strip's rows and this chunk's columns (fixed order: 16 ranges summed in order, combined in order) ----
    if constexpr (!SHIFTED) {
        {
            float s = 0.f;
            if constexpr (SMALL) {
#pragma unroll
                for (int u = 0; u < 8; ++u) s += (pr_r < nrows && pr_tb + u < pr_te) ? xr[u] : 0.f;
            } else if (pr_r < nrows) {
                for (int t0 = pr_tb; t0 < pr_te; t0 += 8) {   // 8 loads in flight at a time, summed in order
                    float x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = t0 + u < pr_te ? pr_src[(size_t)(t0 + u) * L.n1p] : 0.f;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (t0 + u < pr_te) s += x[u];
                }
            }
            red[pr_g][pr_r] = s;
        }
        {
            float4 c = make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (SMALL) {   // one partial per range -> a plain in-order sum
                if (pc_on) {
                    c = xc[0];
#pragma unroll
                    for (int t = 1; t < 16; ++t)
                        if (t < nrt) { c.x += xc[t].x; c.y += xc[t].y; c.z += xc[t].z; c.w += xc[t].w; }
                }
            } else if (pc_on) {
                {
                    for (int r = 0; r < 16; ++r) {
                        const int tb = r * pc_per, te = min(nrt, tb + pc_per);
                        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (int t = tb; t < te; ++t) {
                            const float4 x = *reinterpret_cast<const float4*>(pc_src + (size_t)t * L.n2p);
                            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
                        }
                        if (r == 0) c = s;
                        else { c.x += s.x; c.y += s.y; c.z += s.z; c.w += s.w; }
                    }
                }
            }
            // the element loop multiplies by reciprocal normalisers: one IEEE division per column / row, not two per element
            c.x = 1.f / c.x; c.y = 1.f / c.y; c.z = 1.f / c.z; c.w = 1.f / c.w;
            if (pc_jl < CF_COLS) *reinterpret_cast<float4*>(cs_s + pc_jl) = c;
        }
        __syncthreads();
        if (tid < CF_ROWS) {
            float tot = red[0][tid];
#pragma unroll
            for (int p = 1; p < 16; ++p) tot += red[p][tid];
            rs_s[tid] = 1.f / tot;
        }
    } else {
        for (int jl = tid; jl < CF_COLS; jl += 256) {
            const int j = j0 + jl;
            cs_s[jl] = j < L.n2 ? 1.f / cs_g[(size_t)f * L.n2p + j] : 1.f;
            csh_s[jl] = j < L.n2 ? cshift[(size_t)f * L.n2p + j] : 0.f;
        }
        if (tid < CF_ROWS) {
            const bool ok = tid < nrows;
            rs_s[tid] = ok ? 1.f / rs_g[(size_t)f * L.n1p + i0 + tid] : 1.f;
            rsh_s[tid] = ok ? rshift[(size_t)f * L.n1p + i0 + tid] : 0.f;
        }
    }

    __syncthreads();   // normalisers visible
    float csj[CF_NQ][4], cshj[CF_NQ][4], cmv[CF_NQ][4];
    int cmi[CF_NQ][4];
#pragma unroll
    for (int q = 0; q < CF_NQ; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            csj[q][k] = cs_s[lcol(q, k)];
            cshj[q][k] = SHIFTED ? csh_s[lcol(q, k)] : 0.f;
            cmv[q][k] = -INFINITY;
            cmi[q][k] = 0;
        }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = 4 * wave + u;
        if (r < nrows) {
            const int i = i0 + r;
            const size_t base = (size_t)i * L.n2;
            const float rsi = rs_s[r];
            const float rshi = SHIFTED ? rsh_s[r] : 0.f;
            float rv = -INFINITY;
            int ri = 0x7fffffff;
#pragma unroll
            for (int q = 0; q < CF_NQ; ++q) {
                float c[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int j = j0 + lcol(q, k);
                    if (j < L.n2) {
                        const float x = e[u][q][k];
                        // csj / rsi hold RECIPROCAL sums: softmax(dim=1) * softmax(dim=2) = (E * 1/colsum) * (E * 1/rowsum)
                        if constexpr (SHIFTED) c[k] = (expf(x - cshj[q][k]) * csj[q][k]) * (expf(x - rshi) * rsi);
                        else c[k] = (x * csj[q][k]) * (x * rsi);
                        if (c[k] > cmv[q][k]) { cmv[q][k] = c[k]; cmi[q][k] = i; }
                        if (c[k] > rv) { rv = c[k]; ri = j; }   // a lane visits its columns in increasing order: strict '>' keeps the first
                    }
                }
                if (VEC) {
                    const int j = j0 + lcol(q, 0);
                    if (j < L.n2) *reinterpret_cast<float4*>(cf + base + j) = make_float4(c[0], c[1], c[2], c[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int j = j0 + lcol(q, k);
                        if (j < L.n2) cf[base + j] = c[k];
                    }
                }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const float ov = __shfl_xor(rv, o);
                const int oi = __shfl_xor(ri, o);
                argmax_combine(rv, ri, ov, oi);
            }
            if (lane == 0) {
                rmax_v[((size_t)f * nch + chk) * L.n1p + i] = rv;
                rmax_i[((size_t)f * nch + chk) * L.n1p + i] = ri;
            }
        }
    }
    // ---- column maxima of the strip: the 4 waves (ascending rows) are combined in order, strict '>' keeps the first row ----
#pragma unroll
    for (int q = 0; q < CF_NQ; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cmv_s[wave][lcol(q, k)] = cmv[q][k];
            cmi_s[wave][lcol(q, k)] = cmi[q][k];
        }
    __syncthreads();
    {
        const int jl = 4 * tid, j = j0 + jl;
        if (jl < CF_COLS && j < L.n2p) {
            float4 v = *reinterpret_cast<const float4*>(&cmv_s[0][jl]);
            int4 a = *reinterpret_cast<const int4*>(&cmi_s[0][jl]);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 ov = *reinterpret_cast<const float4*>(&cmv_s[w][jl]);
                const int4 oa = *reinterpret_cast<const int4*>(&cmi_s[w][jl]);
                if (ov.x > v.x) { v.x = ov.x; a.x = oa.x; }
                if (ov.y > v.y) { v.y = ov.y; a.y = oa.y; }
                if (ov.z > v.z) { v.z = ov.z; a.z = oa.z; }
                if (ov.w > v.w) { v.w = ov.w; a.w = oa.w; }
            }
            *reinterpret_cast<float4*>(cmax_v + ((size_t)f * nst + st) * L.n2p + j) = v;
            *reinterpret_cast<int4*>(cmax_i + ((size_t)f * nst + st) * L.n2p + j) = a;
        }
    }
}

// mutual check, threshold, -1 fill (GATs_SuperGlue.py:220-237) straight from the finalize kernel's partials: the row
// (max, arg-max) of query i is the first maximum over its nch chunk partials, the column arg-max of 3D point j the first
// maximum over its nst strip partials (partials are ordered by increasing index; strict '>' keeps the first).
//   row item i:    j = argmax_j conf[i, :];  mutual0 = argmax_i conf[:, j] == i
//   column item j: i = argmax_i conf[:, j];  mutual1 = argmax_j conf[i, :] == j   (which implies mutual0 of i)
// One workgroup = 32 items (rows, or columns) x 8 lanes per item: the strip reduction of a column is split 8 ways
// (ranges in order, combined in order), so the dependent-load chains are nst / 8 long.
constexpr int MT_ITEMS = 32;

__global__ __launch_bounds__(256) void match_tail_kernel(const float* __restrict__ rmax_v, const int* __restrict__ rmax_i,
                                                         const float* __restrict__ cmax_v, const int* __restrict__ cmax_i,
                                                         float thr, int64_t* __restrict__ matches0,
                                                         int64_t* __restrict__ matches1, float* __restrict__ ms0,
                                                         float* __restrict__ ms1, ColLayout L, int nch, int nst) {
    __shared__ float pv[8][MT_ITEMS];
    __shared__ int pi[8][MT_ITEMS];
    __shared__ float rowv[MT_ITEMS];
    __shared__ int rowj[MT_ITEMS];
    const int f = blockIdx.y, tid = threadIdx.x, it = tid & (MT_ITEMS - 1), grp = tid / MT_ITEMS;
    const int nrb = L.n1p / MT_ITEMS;
    const bool rows = (int)blockIdx.x < nrb;
    const int base = (rows ? (int)blockIdx.x : (int)blockIdx.x - nrb) * MT_ITEMS;
    const float* rv = rmax_v + (size_t)f * nch * L.n1p;
    const int* ri = rmax_i + (size_t)f * nch * L.n1p;
    const float* cv = cmax_v + (size_t)f * nst * L.n2p;
    const int* ci = cmax_i + (size_t)f * nst * L.n2p;
    // first maximum over partials [tb, te) spaced `stride` apart, 8 loads in flight at a time
    auto argpart = [&](const float* pvv, const int* pii, size_t stride, int tb, int te, float& v) {
        v = -INFINITY;
        int a = 0;
        for (int t0 = tb; t0 < te; t0 += 8) {
            float x[8];
            int xi[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = t0 + u < te;
                x[u] = in ? pvv[(size_t)(t0 + u) * stride] : -INFINITY;
                xi[u] = in ? pii[(size_t)(t0 + u) * stride] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (x[u] > v) { v = x[u]; a = xi[u]; }
        }
        return a;
    };
    auto rowarg = [&](int i, float& v) { return argpart(rv + i, ri + i, (size_t)L.n1p, 0, nch, v); };
    // first maximum of column j over this lane group's range of strips
    auto colarg_part = [&](int j, float& v) {
        const int per = (nst + 7) / 8;
        const int tb = grp * per;
        return argpart(cv + j, ci + j, (size_t)L.n2p, tb, min(nst, tb + per), v);
    };
    auto colarg_combine = [&]() {   // call with tid < MT_ITEMS after pv / pi are written and synchronised
        float v = pv[0][it];
        int a = pi[0][it];
#pragma unroll
        for (int g = 1; g < 8; ++g)
            if (pv[g][it] > v) { v = pv[g][it]; a = pi[g][it]; }
        return a;
    };
    if (rows) {
        const int i = base + it;
        const bool live = i < L.n1;
        if (grp == 0) {
            float v = 0.f;
            const int j = live ? rowarg(i, v) : 0;
            rowv[it] = v;
            rowj[it] = j;
        }
        __syncthreads();
        float v;
        pi[grp][it] = colarg_part(rowj[it], v);
        pv[grp][it] = v;
        __syncthreads();
        if (grp == 0 && live) {
            const bool mutual0 = colarg_combine() == i;
            const float s0 = mutual0 ? rowv[it] : 0.f;
            const bool valid0 = mutual0 && s0 > thr;
            matches0[(size_t)f * L.n1 + i] = valid0 ? (int64_t)rowj[it] : (int64_t)-1;
            ms0[(size_t)f * L.n1 + i] = s0;
        }
    } else {
        const int j = base + it;
        const bool live = j < L.n2;
        float v;
        pi[grp][it] = colarg_part(live ? j : 0, v);
        pv[grp][it] = v;
        __syncthreads();
        if (grp == 0 && live) {
            const int i = colarg_combine();
            float m;
            const bool mutual1 = rowarg(i, m) == j;
            const float s1 = mutual1 ? m : 0.f;
            const bool valid1 = mutual1 && m > thr;
            matches1[(size_t)f * L.n2 + j] = valid1 ? (int64_t)i : (int64_t)-1;
            ms1[(size_t)f * L.n2 + j] = s1;
        }
    }
}

// ---- max-subtracting softmax statistics of the raw score matrix S (only when 1 / scale_factor > 80; off the benchmarked
//      path, simple kernels): per row i  rshift = max_j S_ij, rs = sum_j exp(S_ij - rshift); per column likewise ----
__global__ __launch_bounds__(256) void softmax_rowstat_kernel(const float* __restrict__ S, float* __restrict__ rshift,
                                                              float* __restrict__ rs, ColLayout L) {
    __shared__ float red[256];
    const int i = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const float* row = S + ((size_t)f * L.n1 + i) * L.n2;
    float m = -INFINITY;
    for (int j = tid; j < L.n2; j += 256) m = fmaxf(m, row[j]);
    red[tid] = m;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
        __syncthreads();
    }
    m = red[0];
    __syncthreads();
    float s = 0.f;
    for (int j = tid; j < L.n2; j += 256) s += expf(row[j] - m);
    red[tid] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {   // fixed tree
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        rshift[(size_t)f * L.n1p + i] = m;
        rs[(size_t)f * L.n1p + i] = red[0];
    }
}

__global__ __launch_bounds__(256) void softmax_colstat_kernel(const float* __restrict__ S, float* __restrict__ cshift,
                                                              float* __restrict__ cs, ColLayout L) {
    const int j = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (j >= L.n2) return;
    const float* col = S + (size_t)f * L.n1 * L.n2 + j;
    float m = -INFINITY;
    for (int i = 0; i < L.n1; ++i) m = fmaxf(m, col[(size_t)i * L.n2]);
    float s = 0.f;
    for (int i = 0; i < L.n1; ++i) s += expf(col[(size_t)i * L.n2] - m);
    cshift[(size_t)f * L.n2p + j] = m;
    cs[(size_t)f * L.n2p + j] = s;
}

void launch_dual_softmax_match(const Workspace& w, float* conf, float scale, int shifted, float thr, int64_t* matches0,
                               int64_t* matches1, float* mscores0, float* mscores1, hipStream_t s, ProfileHook* hk) {
    const ColLayout& L = w.L;
    (void)scale;
    const dim3 gf(w.cf_nch, w.cf_nst, L.b);
    const int nrt = (L.n1p + score_tile_rows() - 1) / score_tile_rows();
    const int nct = L.n2p / score_tile_cols();
    const bool vec = (L.n2 & 3) == 0 && (reinterpret_cast<uintptr_t>(conf) & 15) == 0;
    if (shifted) {
        GATSSPG_LAUNCH(hk, KID_SOFTMAX_STATS, s, softmax_rowstat_kernel, dim3(L.n1, L.b), dim3(256), 0, s, conf, w.rshift, w.rs, L);
        GATSSPG_LAUNCH(hk, KID_SOFTMAX_STATS, s, softmax_colstat_kernel, dim3((L.n2 + 255) / 256, L.b), dim3(256), 0, s, conf,
                       w.cshift, w.cs, L);
    }
    const bool small = nct <= 128 && nrt <= 16;
#define GATSSPG_FINALIZE(VEC_, SH_, SM_)                                                                                     \
    GATSSPG_LAUNCH(hk, KID_CONF_FINALIZE, s, (conf_finalize_kernel<VEC_, SH_, SM_>), gf, dim3(256), 0, s, conf, w.rowpart,    \
                   w.colpart, w.rs, w.cs, w.rshift, w.cshift, w.rmax_v, w.rmax_i, w.cmax_v, w.cmax_i, L, nct, nrt,            \
                   w.cf_nch, w.cf_nst)
    if (vec && !shifted && small) GATSSPG_FINALIZE(true, false, true);
    else if (vec && !shifted) GATSSPG_FINALIZE(true, false, false);
    else if (!shifted && small) GATSSPG_FINALIZE(false, false, true);
    else if (!shifted) GATSSPG_FINALIZE(false, false, false);
    else if (vec) GATSSPG_FINALIZE(true, true, true);
    else GATSSPG_FINALIZE(false, true, true);
#undef GATSSPG_FINALIZE
    const dim3 g1((L.n1p + L.n2p) / MT_ITEMS, L.b);
    GATSSPG_LAUNCH(hk, KID_MATCH_TAIL, s, match_tail_kernel, g1, dim3(256), 0, s, w.rmax_v, w.rmax_i, w.cmax_v, w.cmax_i, thr,
                   matches0, matches1, mscores0, mscores1, L, w.cf_nch, w.cf_nst);
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
