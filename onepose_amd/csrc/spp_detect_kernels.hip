// SuperPoint extractor, detector / descriptor heads after the convolutions
// (reference: src/models/extractors/SuperPoint/superpoint.py:47-94,158-195).
//
// Every step that decides WHICH pixels become keypoints (exact float equality in the NMS, threshold, border,
// top-k) is integer / comparison work on one fp32 score map and is reproduced bit for bit: given the same score
// map the keypoint list is identical to the reference's (tests feed the oracle's map through spp_detect).
#include "spp_common.h"

namespace spp {

// =====================================================================================================
// softmax over the 65 cell channels, dustbin dropped, 8x8 cell shuffle (:158-162)
//   LG [65(+pad)][ldt] padded 1/8-resolution plane -> score [b][H][W]
// =====================================================================================================
__global__ __launch_bounds__(64) void score_map_kernel(const float* __restrict__ LG, FeatLayout Lc, float* __restrict__ score) {
    const int cell = blockIdx.x * 64 + threadIdx.x;
    const int im = blockIdx.y;
    if (cell >= Lc.H * Lc.W) return;
    const int cy = cell / Lc.W, cx = cell - cy * Lc.W;
    const float* src = LG + (size_t)im * Lc.ld + (size_t)(cy + 1) * Lc.Wp + cx + 1;
    float l[65];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 65; ++r) {
        l[r] = src[(size_t)r * Lc.ldt];
        m = fmaxf(m, l[r]);
    }
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 65; ++r) {
        l[r] = expf(l[r] - m);
        sum += l[r];
    }
    const int W = Lc.W * 8;
    float* dst = score + (size_t)im * Lc.H * 8 * W + (size_t)cy * 8 * W + cx * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float4 lo = make_float4(l[i * 8 + 0] / sum, l[i * 8 + 1] / sum, l[i * 8 + 2] / sum, l[i * 8 + 3] / sum);
        float4 hi = make_float4(l[i * 8 + 4] / sum, l[i * 8 + 5] / sum, l[i * 8 + 6] / sum, l[i * 8 + 7] / sum);
        *reinterpret_cast<float4*>(dst + (size_t)i * W) = lo;
        *reinterpret_cast<float4*>(dst + (size_t)i * W + 4) = hi;
    }
}

// =====================================================================================================
// simple_nms (:47-62), all five max-pools of the two suppression rounds inside one workgroup: a T x T output tile
// with a 5*R halo (the dependency radius of the final mask) lives in LDS; max-pools are separable.
// Out-of-image taps are -inf exactly as max_pool2d pads.
// Each pass gives a thread a strip of 4 consecutive elements along the pooled direction (window values in registers:
// 4 + 2R LDS reads for 4 outputs) with the 64 lanes of a wave spread along the other direction; row strides of 65
// floats / 68 bytes make both directions bank-conflict-free.  R is a template parameter so the windows unroll.
// =====================================================================================================
constexpr int E = NMS_E;
constexpr int NMS_THREADS = 1024;
constexpr int FS = E + 1, BS = E + 4;
struct NmsSmem {
    float s[E * FS], ss[E * FS], t[E * FS];
    unsigned char m[E * BS], sup[E * BS], tb[E * BS];
};

// out[p] = op over in[p-R .. p+R] along columns (ALONG_COLS: lane = row, strip walks columns) or rows
template <int R, int S, bool ALONG_COLS, class V, class Op>
__device__ __forceinline__ void pool_strip(const V* __restrict__ in, V* __restrict__ out, V ident, Op op) {
    const int lane = threadIdx.x & 63, p0 = (threadIdx.x >> 6) * 4;
    V v[4 + 2 * R];
#pragma unroll
    for (int j = 0; j < 4 + 2 * R; ++j) {
        const int p = p0 - R + j;
        v[j] = (p >= 0 && p < E) ? in[ALONG_COLS ? lane * S + p : p * S + lane] : ident;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        V acc = v[i];
#pragma unroll
        for (int d = 1; d <= 2 * R; ++d) acc = op(acc, v[i + d]);
        out[ALONG_COLS ? lane * S + p0 + i : (p0 + i) * S + lane] = acc;
    }
}

template <int R>
__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const float* __restrict__ score, float* __restrict__ out, int H, int W,
                                                          int T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_raw[];
    NmsSmem& S = *reinterpret_cast<NmsSmem*>(nms_raw);
    const int im = blockIdx.z;
    const int y0 = blockIdx.y * T - 5 * R, x0 = blockIdx.x * T - 5 * R;   // region origin in the image
    const float* src = score + (size_t)im * H * W;
    const int tid = threadIdx.x;
    const float NINF = -INFINITY;
    auto fmaxop = [](float a, float b) { return fmaxf(a, b); };
    auto orop = [](unsigned char a, unsigned char b) { return (unsigned char)(a | b); };
    // element e of the region: row r = e / E, column c = e % E; four per thread
    bool in_img[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = tid + k * NMS_THREADS, r = e / E, c = e % E;
        const int y = y0 + r, x = x0 + c;
        in_img[k] = y >= 0 && y < H && x >= 0 && x < W;
        S.s[r * FS + c] = in_img[k] ? src[(size_t)y * W + x] : NINF;
    }
    __syncthreads();
    // max_mask = scores == max_pool(scores)                                        (:56)
    pool_strip<R, FS, true>(S.s, S.t, NINF, fmaxop);
    __syncthreads();
    pool_strip<R, FS, false>(S.t, S.ss, NINF, fmaxop);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = tid + k * NMS_THREADS, r = e / E, c = e % E;
        S.m[r * BS + c] = in_img[k] && S.s[r * FS + c] == S.ss[r * FS + c];
    }
    __syncthreads();
    for (int it = 0; it < 2; ++it) {                                                 // (:57-61)
        pool_strip<R, BS, true>(S.m, S.tb, (unsigned char)0, orop);                   // supp_mask = max_pool(max_mask) > 0
        __syncthreads();
        pool_strip<R, BS, false>(S.tb, S.sup, (unsigned char)0, orop);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                                 // supp_scores
            const int e = tid + k * NMS_THREADS, r = e / E, c = e % E;
            S.ss[r * FS + c] = in_img[k] ? (S.sup[r * BS + c] ? 0.f : S.s[r * FS + c]) : NINF;
        }
        __syncthreads();
        pool_strip<R, FS, true>(S.ss, S.t, NINF, fmaxop);                             // max_pool(supp_scores): row pass into S.t
        __syncthreads();
        // column pass straight into registers: this thread owns the four elements (p0 + i, lane) it produces
        float colres[4];
        {
            const int lane = tid & 63, p0 = (tid >> 6) * 4;
            float v[4 + 2 * R];
#pragma unroll
            for (int j = 0; j < 4 + 2 * R; ++j) {
                const int p = p0 - R + j;
                v[j] = (p >= 0 && p < E) ? S.t[p * FS + lane] : NINF;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float acc = v[i];
#pragma unroll
                for (int d = 1; d <= 2 * R; ++d) acc = fmaxf(acc, v[i + d]);
                colres[i] = acc;
            }
            // new_max_mask = supp_scores == pooled; max_mask |= new & ~supp          (element (p0+i, lane))
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = p0 + i, c = lane;
                const int y = y0 + r, x = x0 + c;
                const bool inside = y >= 0 && y < H && x >= 0 && x < W;
                const bool nw = inside && S.ss[r * FS + c] == colres[i];
                S.m[r * BS + c] = S.m[r * BS + c] | (nw && !S.sup[r * BS + c]);
            }
        }
        __syncthreads();
    }
    float* dst = out + (size_t)im * H * W;
    for (int i = tid; i < T * T; i += NMS_THREADS) {
        const int ty = i / T, tx = i % T;
        const int y = blockIdx.y * T + ty, x = blockIdx.x * T + tx;
        if (y < H && x < W) {
            const int r = ty + 5 * R, c = tx + 5 * R;
            dst[(size_t)y * W + x] = S.m[r * BS + c] ? S.s[r * FS + c] : 0.f;         // (:62)
        }
    }
}

// =====================================================================================================
// keypoint list in row-major (torch.nonzero) order: threshold (:165-167) and border (:65-70) tests,
// per-row counts -> exclusive scan -> ordered compaction
// =====================================================================================================
__device__ __forceinline__ bool is_keypoint(float v, int y, int x, int H, int W, float thr, int border) {
    return v > thr && y >= border && y < H - border && x >= border && x < W - border;
}

__global__ __launch_bounds__(64) void rowcount_kernel(const float* __restrict__ nms, int H, int W, float thr, int border,
                                                      int* __restrict__ rowcnt) {
    const int y = blockIdx.x, im = blockIdx.y, lane = threadIdx.x;
    const float* row = nms + ((size_t)im * H + y) * W;
    int cnt = 0;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const bool p = x < W && is_keypoint(row[x < W ? x : 0], y, x, H, W, thr, border);
        cnt += __popcll(__ballot(p));
    }
    if (lane == 0) rowcnt[im * H + y] = cnt;
}

__global__ __launch_bounds__(1024) void rowscan_kernel(const int* __restrict__ rowcnt, int H, int* __restrict__ rowoff,
                                                       int* __restrict__ ncand) {
    __shared__ int part[1024];
    const int im = blockIdx.x, tid = threadIdx.x;
    const int per = (H + 1023) / 1024;
    int local = 0;
    for (int k = 0; k < per; ++k) {
        const int y = tid * per + k;
        if (y < H) local += rowcnt[im * H + y];
    }
    part[tid] = local;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {      // Hillis-Steele inclusive scan (integers: order-independent)
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - local;
    for (int k = 0; k < per; ++k) {
        const int y = tid * per + k;
        if (y < H) {
            rowoff[im * H + y] = run;
            run += rowcnt[im * H + y];
        }
    }
    if (tid == 1023) ncand[im] = part[1023];
}

__global__ __launch_bounds__(64) void compact_kernel(const float* __restrict__ nms, int H, int W, float thr, int border,
                                                     const int* __restrict__ rowoff, int* __restrict__ cand,
                                                     float* __restrict__ cscore) {
    const int y = blockIdx.x, im = blockIdx.y, lane = threadIdx.x;
    const float* row = nms + ((size_t)im * H + y) * W;
    int base = rowoff[im * H + y];
    int* dst = cand + (size_t)im * H * W;
    float* sdst = cscore + (size_t)im * H * W;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const float v = row[x < W ? x : 0];
        const bool p = x < W && is_keypoint(v, y, x, H, W, thr, border);
        const unsigned long long mask = __ballot(p);
        if (p) {
            const int o = base + __popcll(mask & ((1ull << lane) - 1ull));
            dst[o] = y * W + x;
            sdst[o] = v;                       // dense copy of the candidate scores for the top-k passes
        }
        base += __popcll(mask);
    }
}

// =====================================================================================================
// top_k_keypoints (:73-78), engaged when more than k candidates survive:
//   1. select_kernel (one workgroup per image): radix select on the fp32 bit patterns (scores are >= 0, so the
//      unsigned order of the bits is the order of the values) finds the k-th largest score T in three histogram
//      passes (12 + 12 + 8 bits); the k survivors -- score > T, plus the lowest-index candidates among those equal to
//      T -- are compacted in candidate (row-major) order;
//   2. rank_kernel: rank_i = #{j : s_j > s_i or (s_j == s_i and j < i)} among the k survivors, tiled over a fixed grid;
//      partial ranks are integers, so the atomic accumulation is exact and order-independent;
//   3. scatter_kernel: survivor i goes to output slot rank_i (descending score, ties by lower pixel index).
// With n <= k (or k = -1) the row-major list is kept as is.
// =====================================================================================================
__device__ __forceinline__ int block_excl_scan(int v, int* wsum, int& total) {
    // exclusive prefix sum of one int per thread over a 1024-thread workgroup (thread order)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    __syncthreads();
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int t = wsum[w];
        if (w < wave) base += t;
        tot += t;
    }
    total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(1024) void select_kernel(const float* __restrict__ cscore, int HW, const int* __restrict__ ncand,
                                                      const int* __restrict__ cand, int max_kp, int* __restrict__ surv,
                                                      unsigned* __restrict__ skey, int* __restrict__ rank) {
    __shared__ int hist[4096];
    __shared__ int wsum[16];
    __shared__ int sh_bin, sh_need;
    const int im = blockIdx.x, tid = threadIdx.x;
    const int n = ncand[im];
    if (max_kp < 0 || n <= max_kp) return;
    const int* c = cand + (size_t)im * HW;
    const unsigned* key = reinterpret_cast<const unsigned*>(cscore + (size_t)im * HW);   // key[i] = bits of candidate i's score
    // one radix digit: histogram of digit (k >> shift) & (bins-1) over keys whose higher bits equal `prefix`;
    // then the largest digit d with #{digit >= d} >= need.  Returns d, updates need -= #{digit > d}.
    auto digit = [&](unsigned prefix, int pshift, int shift, int bins, int need) {
        for (int i = tid; i < 4096; i += 1024) hist[i] = 0;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const unsigned k = key[i];
            if (pshift >= 32 || (k >> pshift) == prefix) atomicAdd(&hist[(k >> shift) & (bins - 1)], 1);
        }
        __syncthreads();
        // suffix counts: thread t owns bins [4t, 4t+4) (descending scan = exclusive prefix over reversed thread order)
        const int rt = 1023 - tid;                   // reversed owner: thread 0 handles the highest bins
        int loc[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int bin = rt * 4 + (3 - q);        // descending bin order within the thread
            loc[q] = bin < bins ? hist[bin] : 0;
            sum += loc[q];
        }
        int total;
        int above = block_excl_scan(sum, wsum, total);   // keys in bins above this thread's range
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int bin = rt * 4 + (3 - q);
            if (bin < bins && above < need && above + loc[q] >= need) {
                sh_bin = bin;
                sh_need = need - above;
            }
            above += loc[q];
        }
        __syncthreads();
    };
    digit(0u, 32, 20, 4096, max_kp);
    const unsigned b1 = (unsigned)sh_bin;
    const int need1 = sh_need;
    __syncthreads();
    digit(b1, 20, 8, 4096, need1);
    const unsigned b2 = (unsigned)sh_bin;
    const int need2 = sh_need;
    __syncthreads();
    digit((b1 << 12) | b2, 8, 0, 256, need2);
    const unsigned T = (b1 << 20) | (b2 << 8) | (unsigned)sh_bin;
    const int quota = sh_need;                       // candidates equal to T that make the cut (lowest index first)
    __syncthreads();
    // ordered compaction of the k survivors
    int* sv = surv + (size_t)im * HW;
    unsigned* sk = skey + (size_t)im * HW;
    int* rk = rank + (size_t)im * HW;
    // one scan per 1024-candidate chunk: (score > T) in the low half-word, (score == T) in the high one; a tie is kept
    // iff fewer than `quota` ties precede it, so its slot is  #greater before + min(#ties before, quota)
    int gt_before = 0, ties_before = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        const unsigned k = i < n ? key[i] : 0u;
        const int gt = i < n && k > T, tie = i < n && k == T;
        int tot;
        const int pre = block_excl_scan(gt | (tie << 16), wsum, tot);
        const int g = gt_before + (pre & 0xFFFF), t = ties_before + (pre >> 16);
        if (gt || (tie && t < quota)) {
            const int opos = g + min(t, quota);
            sv[opos] = c[i];
            sk[opos] = k;
        }
        gt_before += tot & 0xFFFF;
        ties_before += tot >> 16;
    }
    for (int i = tid; i < max_kp; i += 1024) rk[i] = 0;      // accumulators of rank_kernel
}

constexpr int RK_GX = 64, RK_GY = 16, RK_J = 256;

__global__ __launch_bounds__(256) void rank_kernel(const unsigned* __restrict__ skey, int HW, const int* __restrict__ ncand,
                                                   int max_kp, int* __restrict__ rank) {
    __shared__ unsigned long long sj[RK_J];
    const int im = blockIdx.z, tid = threadIdx.x;
    if (max_kp < 0 || ncand[im] <= max_kp) return;
    const int n = max_kp;                                     // the survivors of select_kernel
    const unsigned* key = skey + (size_t)im * HW;
    // composite key: (score bits, ~position) -> "greater" = higher score, or equal score and lower position
    auto ckey = [&](int i) { return ((unsigned long long)key[i] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i); };
    const int nic = (n + 255) / 256, njc = (n + RK_J - 1) / RK_J;
    for (int ic = blockIdx.x; ic < nic; ic += RK_GX) {
        const int i = ic * 256 + tid;
        const unsigned long long ki = i < n ? ckey(i) : ~0ull;
        int r = 0;
        for (int jc = blockIdx.y; jc < njc; jc += RK_GY) {
            const int j0 = jc * RK_J;
            __syncthreads();
            for (int t = tid; t < RK_J; t += 256) sj[t] = (j0 + t < n) ? ckey(j0 + t) : 0ull;
            __syncthreads();
#pragma unroll 8
            for (int j = 0; j < RK_J; ++j) r += sj[j] > ki;
        }
        if (i < n && r) atomicAdd(rank + (size_t)im * HW + i, r);
    }
}

__global__ __launch_bounds__(256) void scatter_kernel(int HW, const int* __restrict__ ncand, const int* __restrict__ cand,
                                                      const int* __restrict__ surv, const int* __restrict__ rank, int max_kp,
                                                      int capacity, int* __restrict__ sel, int32_t* __restrict__ counts) {
    const int im = blockIdx.y;
    const int n = ncand[im];
    const bool topk = max_kp >= 0 && n > max_kp;
    const int nout = topk ? max_kp : (n < capacity ? n : capacity);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counts[im * 2 + 0] = nout;
        counts[im * 2 + 1] = n;
    }
    const int* src = (topk ? surv : cand) + (size_t)im * HW;
    const int* rk = rank + (size_t)im * HW;
    int* o = sel + (size_t)im * HW;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nout; i += gridDim.x * 256) o[topk ? rk[i] : i] = src[i];
}

// =====================================================================================================
// descriptors: 1 / max(||d||, eps) per cell (:185), then per keypoint bilinear sampling of the normalised
// map and a second normalisation (:81-94).
// =====================================================================================================
__global__ __launch_bounds__(256) void cellnorm_kernel(DescView dv, int Hc, int Wc, float* __restrict__ invn) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int cell = blockIdx.x * 64 + cl, im = blockIdx.y;
    float ss = 0.f;
    if (cell < Hc * Wc) {
        const int cy = cell / Wc, cx = cell - cy * Wc;
        const float* p = dv.p + (size_t)im * dv.istride + dv.origin + (size_t)cy * dv.rstride + (size_t)cx * dv.xstride + (size_t)(grp * 64) * dv.cstride;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) {
            const float v = p[(size_t)c * dv.cstride];
            ss = fmaf(v, v, ss);
        }
    }
    part[grp][cl] = ss;
    __syncthreads();
    if (grp == 0 && cell < Hc * Wc) {
        const float tot = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        invn[(size_t)im * Hc * Wc + cell] = 1.f / fmaxf(sqrtf(tot), 1e-12f);
    }
}

// position-major descriptors (cstride 1): one wave per cell, a lane owns 4 consecutive channels (one 16-byte load), 4 cells per workgroup
__global__ __launch_bounds__(256) void cellnorm_pm_kernel(DescView dv, int Hc, int Wc, float* __restrict__ invn) {
    const int lane = threadIdx.x & 63, cell = blockIdx.x * 4 + (threadIdx.x >> 6), im = blockIdx.y;
    if (cell >= Hc * Wc) return;
    const int cy = cell / Wc, cx = cell - cy * Wc;
    const float4 v = *reinterpret_cast<const float4*>(dv.p + (size_t)im * dv.istride + dv.origin + (size_t)cy * dv.rstride + (size_t)cx * dv.xstride + 4 * lane);
    float ss = fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, v.x * v.x)));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (lane == 0) invn[(size_t)im * Hc * Wc + cell] = 1.f / fmaxf(sqrtf(ss), 1e-12f);
}

// one workgroup = 16 keypoints x 256 channels (16 channel groups of 16)
__global__ __launch_bounds__(256) void sample_kernel(DescView dv, int Hc, int Wc, const float* __restrict__ invn,
                                                     const float* __restrict__ nms, const int* __restrict__ sel,
                                                     const int32_t* __restrict__ counts, int H, int W, int align_corners,
                                                     int capacity, float* __restrict__ keypoints, float* __restrict__ scores,
                                                     float* __restrict__ desc) {
    __shared__ float part[16][16];
    const int im = blockIdx.y, tid = threadIdx.x;
    const int kl = tid & 15, grp = tid >> 4;
    const int slot = blockIdx.x * 16 + kl;
    const int nout = counts[im * 2];
    if (blockIdx.x * 16 >= nout) return;
    const bool live = slot < nout;
    int pix = 0;
    if (live) pix = sel[(size_t)im * H * W + slot];
    const int py = pix / W, px = pix - py * W;
    // sample_descriptors (:83-88) with s = 8: keypoints - s/2 + 0.5, / (w*s - s/2 - 0.5), *2 - 1
    const float kx = (float)px - 4.f + 0.5f, ky = (float)py - 4.f + 0.5f;
    const float gx = kx / ((float)(Wc * 8) - 4.f - 0.5f) * 2.f - 1.f;
    const float gy = ky / ((float)(Hc * 8) - 4.f - 0.5f) * 2.f - 1.f;
    float ix, iy;
    if (align_corners) {
        ix = (gx + 1.f) / 2.f * (float)(Wc - 1);
        iy = (gy + 1.f) / 2.f * (float)(Hc - 1);
    } else {
        ix = ((gx + 1.f) * (float)Wc - 1.f) / 2.f;
        iy = ((gy + 1.f) * (float)Hc - 1.f) / 2.f;
    }
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    float wq[4];
    int off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int xi = x0 + (q & 1), yi = y0 + (q >> 1);
        const bool ok = live && xi >= 0 && xi < Wc && yi >= 0 && yi < Hc;
        const float w = (1.f - fabsf(ix - (float)xi)) * (1.f - fabsf(iy - (float)yi));
        const int cell = ok ? yi * Wc + xi : 0;
        wq[q] = ok ? w * invn[(size_t)im * Hc * Wc + cell] : 0.f;
        off[q] = ok ? yi * dv.rstride + xi * dv.xstride : 0;
    }
    const float* base = dv.p + (size_t)im * dv.istride + dv.origin;
    float v[16];
    float ss = 0.f;
    // 8 channels x 4 taps = 32 gathers requested before any is used (written as one fmaf chain per channel hipcc waited for
    // every gather -- s_waitcnt vmcnt(0) -- before issuing the next: 64 dependent L2 round trips per thread)
    if (dv.cstride == 1) {
        // position-major plane (the extractor's own): this thread's 16 channels of a tap are 64 contiguous bytes -- 16 float4
        // loads instead of 64 scattered dwords (the channel-major gathers touch 1024 cache lines per keypoint, this form 16)
        float4 u[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) u[q][j] = *reinterpret_cast<const float4*>(base + off[q] + grp * 16 + 4 * j);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float tq[4][4] = {{u[0][j].x, u[1][j].x, u[2][j].x, u[3][j].x}, {u[0][j].y, u[1][j].y, u[2][j].y, u[3][j].y},
                                    {u[0][j].z, u[1][j].z, u[2][j].z, u[3][j].z}, {u[0][j].w, u[1][j].w, u[2][j].w, u[3][j].w}};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) a = fmaf(tq[c][q], wq[q], a);
                v[4 * j + c] = a;
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) ss = fmaf(v[k], v[k], ss);
    } else {
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 8) {
            float t[8][4];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float* pc = base + (size_t)(grp * 16 + k0 + k) * dv.cstride;
#pragma unroll
                for (int q = 0; q < 4; ++q) t[k][q] = pc[off[q]];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) a = fmaf(t[k][q], wq[q], a);
                v[k0 + k] = a;
                ss = fmaf(a, a, ss);
            }
        }
    }
    part[grp][kl] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) tot += part[g][kl];
    const float inv = 1.f / fmaxf(sqrtf(tot), 1e-12f);
    if (!live) return;
    float* d = desc + (size_t)im * DD * capacity + slot;
#pragma unroll
    for (int k = 0; k < 16; ++k) d[(size_t)(grp * 16 + k) * capacity] = v[k] * inv;
    if (grp == 0) {
        keypoints[((size_t)im * capacity + slot) * 2 + 0] = (float)px;     // (:180) (h, w) -> (x, y)
        keypoints[((size_t)im * capacity + slot) * 2 + 1] = (float)py;
        scores[(size_t)im * capacity + slot] = nms[(size_t)im * H * W + pix];
    }
}

// =====================================================================================================
// launchers
// =====================================================================================================
void launch_score_map(const Workspace& w, float* score_map, hipStream_t s, ProfileHook* hk) {
    const int cells = w.L4.H * w.L4.W;
    SPP_LAUNCH(hk, KID_SCORE, s, score_map_kernel, dim3((cells + 63) / 64, w.L4.b), dim3(64), 0, s, w.lg, w.L4, score_map);
}

void launch_detect(const float* score_map, DescView dv, const Workspace& w, const DetectParams& dp, float* keypoints,
                   float* scores, float* descriptors, int32_t* counts, float* nms_out, hipStream_t s, ProfileHook* hk) {
    // the score map covers the (H/8)*8 x (W/8)*8 top-left part of the image (:160-162: h*8, w*8 after three floor poolings)
    const int b = w.L1.b, Hc = w.L4.H, Wc = w.L4.W, H = Hc * 8, W = Wc * 8;
    float* nms = nms_out ? nms_out : w.nms;
    const int R = dp.nms_radius;
    if (R == 0) {
        (void)hipMemcpyAsync(nms, score_map, sizeof(float) * (size_t)b * H * W, hipMemcpyDeviceToDevice, s);   // :62 with an all-true mask
    } else {
        const int T = E - 10 * R < 32 ? E - 10 * R : 32;
        const dim3 grid((W + T - 1) / T, (H + T - 1) / T, b);
        auto go = [&](auto kern) {
            static bool attr_done[16] = {};            // one flag set per instantiation (this lambda is instantiated per radius)
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (dev < 16 && !attr_done[dev]) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)sizeof(NmsSmem));
                attr_done[dev] = true;
            }
            SPP_LAUNCH(hk, KID_NMS, s, kern, grid, dim3(NMS_THREADS), sizeof(NmsSmem), s, score_map, nms, H, W, T);
        };
        switch (R) {
            case 1: go(nms_kernel<1>); break;
            case 2: go(nms_kernel<2>); break;
            case 3: go(nms_kernel<3>); break;
            case 4: go(nms_kernel<4>); break;
            case 5: go(nms_kernel<5>); break;
            default: go(nms_kernel<6>); break;
        }
    }
    SPP_LAUNCH(hk, KID_ROWCOUNT, s, rowcount_kernel, dim3(H, b), dim3(64), 0, s, nms, H, W, dp.threshold, dp.remove_borders,
               w.rowcnt);
    SPP_LAUNCH(hk, KID_SCAN, s, rowscan_kernel, dim3(b), dim3(1024), 0, s, w.rowcnt, H, w.rowoff, w.ncand);
    SPP_LAUNCH(hk, KID_COMPACT, s, compact_kernel, dim3(H, b), dim3(64), 0, s, nms, H, W, dp.threshold, dp.remove_borders,
               w.rowoff, w.cand, w.cscore);
    if (dp.max_keypoints >= 0) {
        SPP_LAUNCH(hk, KID_SELECT, s, select_kernel, dim3(b), dim3(1024), 0, s, w.cscore, H * W, w.ncand, w.cand,
                   dp.max_keypoints, w.surv, w.skey, w.rank);
        SPP_LAUNCH(hk, KID_RANK, s, rank_kernel, dim3(RK_GX, RK_GY, b), dim3(256), 0, s, w.skey, H * W, w.ncand,
                   dp.max_keypoints, w.rank);
    }
    SPP_LAUNCH(hk, KID_SCATTER, s, scatter_kernel, dim3(16, b), dim3(256), 0, s, H * W, w.ncand, w.cand, w.surv, w.rank,
               dp.max_keypoints, dp.capacity, w.sel, counts);
    if (dv.cstride == 1 && dv.xstride % 4 == 0)
        SPP_LAUNCH(hk, KID_CELLNORM, s, cellnorm_pm_kernel, dim3((Hc * Wc + 3) / 4, b), dim3(256), 0, s, dv, Hc, Wc, w.invn);
    else
        SPP_LAUNCH(hk, KID_CELLNORM, s, cellnorm_kernel, dim3((Hc * Wc + 63) / 64, b), dim3(256), 0, s, dv, Hc, Wc, w.invn);
    SPP_LAUNCH(hk, KID_SAMPLE, s, sample_kernel, dim3((dp.capacity + 15) / 16, b), dim3(256), 0, s, dv, Hc, Wc, w.invn, nms,
               w.sel, counts, H, W, dp.align_corners, dp.capacity, keypoints, scores, descriptors);
}

}  // namespace spp
