// MFMA kernels of the GATsSPG forward: everything that is a per-point 1x1-convolution GEMM, the
// linear-attention KV reduction / apply, and the N_2D x N_3D score contraction.
// Reference maths: GATs_SuperGlue.py:69-128 (linear_attention, MultiHeadedAttention,
// AttentionPropagation, MLP) and :209-218 (final_proj, normalize, score einsum, exp of the softmax).
#include "gemm_f32_mfma.h"
#include "gatsspg_launch.h"

namespace gatsspg {

// dynamic LDS of a main loop (operand stages; the epilogues re-use it)
template <class T, int PREC = 0>
constexpr size_t smem_bytes() {
    size_t b = sizeof(float) * T::SMEM_FLOATS;
    if constexpr (PREC == 1 || PREC >= 3) {
        if (Bf3Layout<T>::SMEM_BYTES > b) b = Bf3Layout<T>::SMEM_BYTES;
    }
    if constexpr (PREC == 2) {
        if (Bf6Layout<T>::SMEM_BYTES > b) b = Bf6Layout<T>::SMEM_BYTES;
    }
    return b;
}
template <class T, int PREC>
constexpr int smem_floats_mainloop() { return (int)(smem_bytes<T, PREC>() / sizeof(float)); }

// =====================================================================================================
// K1  QKV projection fused with the linear-attention KV / ksum partial reduction.
//     rows 0..255  : Q = elu(Wq x + bq) + 1  (head-major), written to Qbuf
//     rows 256..767: per head h a 128-row tile [K_h ; V_h]; K = elu(.)+1, V raw.  K and V are
//                    never written to HBM: the tile goes to LDS and a second MFMA pass produces
//                    this column tile's partial  KV_h[q][d] = sum_m V[q][m] K[d][m]  (stored transposed, [d][q]:
//                    kv_final_kernel owns blocks of d rows),  ksum_h[d].
//     (GATs_SuperGlue.py:96-99 projections, :71-72 feature map, :77-78 KV and key.sum)
// =====================================================================================================
using QkvTileW8 = GemmTile<128, QKV_BN, 4, 2, false>;     // both arithmetics: 8 waves, one 32x32 MFMA tile each (fp32: 38.0 vs 40.1 us on 4 waves)
using QkvTileB = GemmTile<128, QKV_BN, 2, 2, false>;      // split-bf16 alternative (tuning builds): 4 waves, 64x32 per wave (31.2 vs 24.6 us)

// PREC = 0: exact fp32 MFMA.  PREC = 1: split-bf16 main loop on the pre-split weight planes Whi / Wlo.
// (forcing 80 VGPRs so that three 8-wave workgroups fit a CU -- the 756 tiles of the headline shape then fit 768 slots in one
// round -- was measured: kernel -2 %, frames/s in flight unchanged; not kept)
// DS: the Q tiles leave straight from the accumulators (store_tile_regs) instead of through an LDS staging tile
// QF: quarter-fragment main loop (gemm_f32_mfma.h; fp32 arithmetic only)
template <class T, int PREC = 0, int BT = 0, int DS = 0, int QF = 0>
__global__ __launch_bounds__(T::THREADS, (PREC >= 2 ? 4 : 1)) void qkv_kv_kernel(const float* __restrict__ Wqkv, const float* __restrict__ bqkv,
                                                            const unsigned short* __restrict__ Whi,
                                                            const unsigned short* __restrict__ Wlo,
                                                            const unsigned short* __restrict__ Wl2,
                                                            const float* __restrict__ Z, float* __restrict__ Qbuf,
                                                            float* __restrict__ kvpart, ColLayout L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (PREC >= 3) fp16_saturate_mode();
    int rt, ct;
    if (!xcd_tile_map(6, active_tiles(L), rt, ct)) return;
    ct = global_tile(L, ct);
    const int c0 = ct * T::BN;
    const int ld = L.ld;
    const float* A = Wqkv + (size_t)rt * 128 * D;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
    // this lane's 16 bias values per MFMA tile (rows 8k + 4 half + 0..3 of the tile: four 16-byte loads), requested BEFORE the
    // main loop.  (Read in the epilogue next to elu's branch they became 16 dependent load -> wait -> write rounds per lane.)
    // (the six-term loop has no 16 registers to park it in and fetches it after the loop)
    float bias[T::TM][16];
    auto load_bias = [&]() {
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const vf4 b4 = ldg4(bqkv + rt * 128 + (wm * T::TM + tm) * 32 + 8 * k + 4 * half);
                bias[tm][4 * k + 0] = b4[0]; bias[tm][4 * k + 1] = b4[1]; bias[tm][4 * k + 2] = b4[2]; bias[tm][4 * k + 3] = b4[3];
            }
    };
    // BT: the bias through an LDS table behind the operand buffers instead (read_bias16, gemm_f32_mfma.h)
    float* btab = smem + smem_floats_mainloop<T, PREC>();
    if constexpr (BT) {
        static_assert(T::BM == 128 && T::KS == 1, "half a piece of bias values, one wave group");
        if ((tid >> 6) == 0 && lane < 32) glds16(bqkv + rt * 128 + 4 * lane, btab);
    } else if constexpr (PREC != 2) {
        load_bias();
    }
    f32x16 acc[T::TM][T::TN];
    zero_acc(acc);
    if constexpr (PREC == 1 || PREC >= 3) {   // PREC >= 3: the planes hold fp16 terms, the products run on the f16 MFMA (4: four products)
        // weight planes are slab-major ([K/32][rows][32], split_weights_kernel): a 128 x 32 slab is 8 KB of consecutive bytes
        const size_t ro = (size_t)rt * 128 * BK;
        auto ah = [&](int kt) { return Whi + ro + (size_t)kt * 768 * BK; };
        auto alo = [&](int kt) { return Wlo + ro + (size_t)kt * 768 * BK; };
        auto bl = [&](int kt) { return Z + (size_t)kt * BK * ld + c0; };
        gemm_mainloop_bf3<T, decltype(ah), decltype(alo), decltype(bl), NoHooks, (PREC >= 3), (PREC == 4 ? 4 : 3)>(
            acc, reinterpret_cast<unsigned short*>(smem), D / BK, ah, alo, BK, bl, ld);
    } else if constexpr (PREC == 2) {
        const size_t ro = (size_t)rt * 128 * BK;
        gemm_mainloop_bf6<T>(
            acc, reinterpret_cast<unsigned short*>(smem), D / BK,
            [&](int kt, int pl) { return (pl == 0 ? Whi : pl == 1 ? Wlo : Wl2) + ro + (size_t)kt * 768 * BK; }, BK,
            [&](int kt) { return Z + (size_t)kt * BK * ld + c0; }, ld);
    } else {
        auto al = [&](int kt) { return A + kt * BK; };
        auto bl = [&](int kt) { return Z + (size_t)kt * BK * ld + c0; };
        gemm_mainloop<T, decltype(al), decltype(bl), 0, IdentityCol, NoHooks, QF>(acc, smem, D / BK, al, D, bl, ld);
    }
    if constexpr (BT) read_bias16<T>(btab, wm, half, bias);
    else if constexpr (PREC == 2) load_bias();

    if (rt < 2) {
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][0][r] = elu1_select(acc[tm][0][r] + bias[tm][r]) + 1.f;
        if constexpr (DS) store_tile_regs<T>(acc, Qbuf + (size_t)rt * 128 * ld + c0, ld, [](int, float v) { return v; });
        else store_tile_via_lds<T>(acc, smem, Qbuf + (size_t)rt * 128 * ld + c0, ld, [](int, float v) { return v; });
        return;
    }
    // ---- K_h / V_h tile -> LDS -> KV partial ----
    const int h = rt - 2;
    const TileSeg ts = tile_seg(L, c0, T::BN);
    constexpr int TS = T::BN + 4;  // LDS row stride (floats) of the [128][64] K/V tile: b128-read conflict-free
    float* Tl = smem;              // 128 * 68 floats = 34.8 KB <= main-loop LDS (reusable after its last barrier)
    float opmx = 0.f;              // largest K (> 0) or |V| entry this wave holds: its rows are all K_h or all V_h
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * T::TM + tm) * 32 + mfma_row(r, half);  // 0..63 = K_h channel d, 64..127 = V_h channel q
            const int col = wn * 32 + l31;
            float v = acc[tm][0][r] + bias[tm][r];
            const float kf = elu1_select(v) + 1.f;
            v = row < 64 ? kf : v;
            v = col >= ts.valid ? 0.f : v;  // pad columns must not enter the sums (elu(0)+1 = 1)
            opmx = fmaxf(opmx, fabsf(v));
            Tl[row * TS + col] = v;
        }
    {   // bound data of the tile for the fp16 modes' message-operator scale (kv_final_kernel): slots [4..7] = max |V| of the tile, one per wave
        // that holds V_h rows (the other V slots 0); slots [0..3] = an upper bound of the tile's key sums per K-holding wave.  Written in EVERY
        // arithmetic: a KV partial / database cache must be valid input for the fp16 modes whatever flags it was prepared under (the C ABI
        // does not tie a cache to them: round-5 advisor, medium) -- never uninitialised workspace
        static_assert(T::WAVES_MN * T::KS == 8 || (T::WAVES_MN * T::KS == 4 && T::TM == 2), "8 waves: 0..3 hold K_h, 4..7 V_h; 4 waves: 0, 1 K_h, 2, 3 V_h");
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) opmx = fmaxf(opmx, __shfl_xor(opmx, o));
        // Slots 0..3 here: 64 * (largest K of the wave's quadrant) -- a cheaper, looser member of the same family of bounds (a tile's key sum
        // over 64 columns is at most 64 max K; the average of two quadrants' bounds is at most the larger): this kernel's arithmetics never
        // read the slots themselves, they only have to be VALID for an fp16-mode consumer of a cache prepared here, and the tight form (the
        // per-wave largest key sum, qkv_kv_sp_kernel) measured +2.4 us on this kernel (87 instead of 80 registers: two workgroups per CU).
        if (lane == 0) {
            float* mx = kvpart + ((size_t)ct * H + h) * KVP + DH * DH + DH;
            if constexpr (T::WAVES_MN * T::KS == 8) {
                mx[wave] = wave < 4 ? 64.f * opmx : opmx;
            } else {
                mx[wave] = wm == 0 ? 64.f * opmx : 0.f;
                mx[4 + wave] = wm == 0 ? 0.f : opmx;
            }
        }
    }
    __syncthreads();
    if (wave < 4) {   // (an 8-wave workgroup leaves this short pass to its first four waves: same order of operations)
        // wave -> 32x32 quadrant (di, qi) of KV^T[d][q]; contraction over the 64 columns m (A operand = K rows, B operand = V rows)
        const int qi = wave >> 1, di = wave & 1;
        f32x16 kv;
#pragma unroll
        for (int r = 0; r < 16; ++r) kv[r] = 0.f;
        const float4* ap = reinterpret_cast<const float4*>(Tl + (di * 32 + l31) * TS + half * 32);
        const float4* bp = reinterpret_cast<const float4*>(Tl + (64 + qi * 32 + l31) * TS + half * 32);
#pragma unroll
        for (int v4 = 0; v4 < 8; ++v4) {
            const float4 a = ap[v4], b = bp[v4];
            kv = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, kv, 0, 0, 0);
            kv = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, kv, 0, 0, 0);
            kv = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, kv, 0, 0, 0);
            kv = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, kv, 0, 0, 0);
        }
        float* out = kvpart + ((size_t)ct * H + h) * KVP;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d = di * 32 + mfma_row(r, half);
            out[d * DH + qi * 32 + l31] = kv[r];
        }
        {   // ksum[d] = sum_m K[d][m]: 4 lanes per row (16 columns each, fixed order), combined by 2 shuffles
            const int d = tid >> 2, qtr = tid & 3;
            const float* kr = Tl + d * TS + qtr * 16;
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 16; ++m) s += kr[m];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            if (qtr == 0) out[DH * DH + d] = s;
        }
    }
}

// K2  fixed-order sum of the KV partials of each (SOURCE segment, head) -> KV^T[seg][h][d][q], ksum[seg][h][d], and -- what
//     replaces the former attn_apply launch and the MSG round trip -- the message operator of the TARGET segment
//     (the same segment for 'self', the other side of the same frame for 'cross', GATs_SuperGlue.py:57,62):
//         M_t[:, h*64 + d] = sum_q (W0b Wm)[:, h*64 + q] KV_h[q][d]            [512 x 256]
//     so that mlp.0's message half  (W0b Wm) msg,  msg_h[q][n] = z_h[n] sum_d KV_h[q][d] Qf_h[d][n]  (:78-79,101,122),
//     becomes  sum_h z_h[n] (M_h Qf_h)[.][n]  inside mlp0_kernel's K loop (AttnFoldHooks).
//     1024 threads = 64 float4 elements x 16 tile-ranges: every range is summed in tile order by one wave
//     (8 independent 16-byte loads in flight at a time on clamped addresses -- a plain unrolled loop leaves a serial
//     remainder loop, one load and one s_waitcnt vmcnt(0) per iteration), the 16 range sums are combined in range
//     order -> the result does not depend on scheduling.  A block owns 4 d rows (all 64 q) of one head: exactly what the
//     4 columns of M it then produces need -- no cross-workgroup dependency.  The weights of the output rows are requested
//     BEFORE the reduction (independent of it).  The operator phase is LDS-bound (every KV value a thread multiplies comes
//     through a ds_read: rows x 1 KB per block), so a thread takes TWO rows per KV read and two workgroups share the rows of a
//     d block (each repeats the reduction).  The last block of a (segment, head) owns ksum.
//     kv_src (amortised mode): final KV sums of the 3D-side sources from the database cache; they are read as a single
//     partial (0 + x = x: same bits as reducing the partials again).
constexpr int KVP4 = KVP / 4;
static_assert(KVP % 4 == 0, "KV partials are summed as float4");
// KVF_RS: row parts of M_t per d block.  2: two workgroups repeat the (cheap, parallel) reduction and each turns it into 256 rows of the
// operator with its first 512 threads (264 workgroups, every CU busy).  1: one workgroup per d block, all 1024 threads in the operator
// phase, every partial read once (136 workgroups, half the bytes).

// abl (tuning builds only, wrong results, 0 in the product): bit 0 no FMA phase, bit 1 no operator stores, bit 2 no weight loads
template <int KVF_RS>
__global__ __launch_bounds__(1024) void kv_final_kernel(const float* __restrict__ kvpart, const float* __restrict__ kv_src,
                                                        float* __restrict__ kvfin, const float* __restrict__ W0,
                                                        float* __restrict__ Mop, unsigned short* __restrict__ Mpl,
                                                        float* __restrict__ ksumT, float* __restrict__ zsc, int* __restrict__ statcnt,
                                                        const float* __restrict__ sc, ColLayout L, int cross, int prec, int abl) {
    __shared__ float4 red[16][64];
    __shared__ float4 kvs[64];   // this block's final KV^T rows: [4 d][16 float4 of q]
    __shared__ float opsum[16];      // per wave: sum over its tiles of the tile's largest key-sum bound (bound data, slots 0..3)
    __shared__ float opmax[16];      // per wave: max |V| over the source segment's tiles (slots 4..7)
    constexpr int KVF_ROWS = 512 / KVF_RS;
    if (prec >= 3) fp16_saturate_mode();
    const int tid = threadIdx.x;
    const int el = tid & 63, part = tid >> 6;
    const bool ksum_block = blockIdx.x == 16 * KVF_RS;
    const int db = ksum_block ? 16 : blockIdx.x / KVF_RS, rs = blockIdx.x % KVF_RS;
    const int e4 = min(db * 64 + el, KVP4 - 1);   // clamped: the last block's spare lanes redo element KVP4 - 1
    const int seg = blockIdx.y / H, h = blockIdx.y % H;
    const int frame = seg >> 1, side = seg & 1;
    if (!((L.side_mask >> side) & 1)) return;
    const int tseg = cross ? (seg ^ 1) : seg;
    // fp16 modes: the operator planes of head h hold sM_h * M_h with a power of two sM_h chosen from a RIGOROUS bound of the operator's
    // entries (round-4 advisor: the former 2^-(ceil(log2 n_src) + 6) heuristic never looked at the data and could saturate silently):
    //     |M_h[r][d]| = |sum_q (W0b Wm)[r][h, q] KV_h[q][d]| <= l1_h * max |KV_h|,
    //     |KV_h[q][d]| = |sum_m V[q][m] K[d][m]| <= vmax_h * sum_m K[d][m] = vmax_h * ksum_h[d]       (K = elu + 1 > 0),
    //     max_d ksum_h[d] <= sum_tiles max_d ksum_tile[d] <= sum_tiles max_w slot_w(tile) =: ksb_h
    // l1_h = largest row-L1 norm of head h's message half (pack time, AttnW::SC[4 + h]); vmax_h = largest |V| entry of the source segment;
    // both data terms ride in the KV partials (every projection kernel writes them: slots 0..3 the per-wave largest key sum of the tile,
    // slots 4..7 max |V|).  Every block of a (segment, head) reduces them with the same loads in the same order: the same scale in all.
    // (Round 6, advisor: the round-5 form bounded KV by n_src * kmax_h * vmax_h -- one outlier K entry lowered the scale of the whole head;
    // a sum of per-tile key-sum maxima moves by that entry / n_src, and is 2-4x tighter on ordinary data.)  sM_h = 2^(14 - e), 2^e > bound:
    // no entry reaches 2^14 (fp16 maximum 65504).  The bound stays loose by the random-sign cancellation of the sums over q and m
    // (typically 2^7 .. 2^12): the largest entries then sit around 2^2 .. 2^7, where both fp16 terms are normal numbers (2^-22 relative).
    // kvfin (and through it the database cache) stores the REDUCED slots -- (ksb, 0, 0, 0) and (vmax, 0, 0, 0) -- so that a cached source, read
    // as a single partial, yields bit for bit the scale of the plain forward (0 + x and max(0, x) are exact).
    // mlp0 folds head h with z_h * zsc_h, zsc_h = (W0 scale) / sM_h: exact.
    const int nsrc = side ? L.n2 : L.n1;
    // operator phase: thread = (row pair rp, q quarter qq); lane qq takes the float4s qq, qq + 4, qq + 8, qq + 12 of the 64 q of a
    // row (the 4 lanes of a row read 64 contiguous bytes per load).  The weights do not depend on the reduction: requested first.
    const int rp = (tid >> 2) & (KVF_ROWS / 2 - 1), qq = tid & 3;
    const int mr = rs * KVF_ROWS + 2 * rp;
    vf4 wv[2][4];
    const bool op_thread = !ksum_block && tid < 2 * KVF_ROWS;
    if (op_thread && !(abl & 4)) {
        const float* wr = W0 + (size_t)mr * 512 + 256 + h * DH + 4 * qq;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[e][i] = ldg4(wr + e * 512 + 16 * i);
    } else {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i) wv[e][i] = (vf4){1.f, 2.f, 3.f, 4.f};
    }
    const bool cached = kv_src != nullptr && side == 1;
    const float* base = cached ? kv_src + ((size_t)frame * H + h) * KVP
                               : kvpart + ((size_t)((frame * L.np + (side ? L.n1p : 0)) / QKV_BN) * H + h) * KVP;
    const int nt = cached ? 1 : (side ? L.n2p : L.n1p) / QKV_BN;
    const int per = (nt + 15) / 16;
    const int tb = part * per, te = min(nt, tb + per);
    // operand maxima of the source's tiles (fp16 modes; requested in front of the reduction's loads, consumed behind them)
    constexpr int MAX4 = (DH * DH + DH) / 4;   // float4 index of a partial's maxima: [max K x 4 waves][max |V| x 4 waves]
    // the bound data of the source's tiles (requested in front of the reduction's loads, consumed behind them); computed in EVERY arithmetic:
    // the ksum block stores the reduced slots with kvfin, and a cache prepared in fp32 must serve the fp16 modes
    float ksb = 0.f, vmx = 0.f;   // ksb: sum over the tiles of the tile's largest key-sum bound (>= max_w sum_tiles slot_w >= max_d ksum[d])
    const bool need_bound = prec >= 3 || ksum_block;   // (block-uniform) fp32 / bf16 launches: only the block that stores kvfin reduces it
    if (need_bound)
    for (int t = tid; t < nt; t += 1024) {
        const float4* mp = reinterpret_cast<const float4*>(base + (size_t)t * H * KVP) + MAX4;
        const float4 a = mp[0], b4 = mp[1];
        ksb += fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w));
        vmx = fmaxf(vmx, fmaxf(fmaxf(b4.x, b4.y), fmaxf(b4.z, b4.w)));
    }
    const bool ismax = e4 >= MAX4;   // (the two bound elements of the partials go through the loop below as well; their result is not used)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int tt = tb; tt < te; tt += 8) {
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            x[u] = *reinterpret_cast<const float4*>(base + (size_t)min(tt + u, nt - 1) * H * KVP + 4 * e4);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (tt + u < te) {
                s.x = ismax ? fmaxf(s.x, x[u].x) : s.x + x[u].x; s.y = ismax ? fmaxf(s.y, x[u].y) : s.y + x[u].y;
                s.z = ismax ? fmaxf(s.z, x[u].z) : s.z + x[u].z; s.w = ismax ? fmaxf(s.w, x[u].w) : s.w + x[u].w;
            }
    }
    red[part][el] = s;
    if (need_bound) {   // fixed-order tree: lanes of a wave by xor-shuffles, then the 16 waves in order (identical in every block of the (segment, head))
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ksb += __shfl_xor(ksb, o);
            vmx = fmaxf(vmx, __shfl_xor(vmx, o));
        }
        if (el == 0) { opsum[part] = ksb; opmax[part] = vmx; }
    }
    __syncthreads();
    if (need_bound) {
        ksb = opsum[0];
        vmx = opmax[0];
#pragma unroll
        for (int p = 1; p < 16; ++p) {
            ksb += opsum[p];
            vmx = fmaxf(vmx, opmax[p]);
        }
    }
    float mscale = 1.f, zfold = 1.f;
    if (prec >= 3) {
        (void)nsrc;
        const float bound = sc[4 + h] * ksb * vmx;
        int e = 0;
        if (bound > 0.f && bound < 3.0e38f) e = min(max(ilogbf(bound) + 1, -100), 100);   // 2^e > bound
        mscale = __builtin_ldexpf(1.f, 14 - e);
        zfold = sc[1] / mscale;
    }
    if (part == 0) {
        float4 tot = red[0][el];
#pragma unroll
        for (int p = 1; p < 16; ++p) {
            const float4 r = red[p][el];
            tot.x = ismax ? fmaxf(tot.x, r.x) : tot.x + r.x; tot.y = ismax ? fmaxf(tot.y, r.y) : tot.y + r.y;
            tot.z = ismax ? fmaxf(tot.z, r.z) : tot.z + r.z; tot.w = ismax ? fmaxf(tot.w, r.w) : tot.w + r.w;
        }
        kvs[el] = tot;
        if (db * 64 + el < KVP4 && rs == 0) {
            if (e4 == MAX4) tot = make_float4(ksb, 0.f, 0.f, 0.f);             // slots 0..3: the tree sum the scale was taken from (max(x, 0, 0, 0) = x)
            else if (e4 == MAX4 + 1) tot = make_float4(vmx, 0.f, 0.f, 0.f);    // slots 4..7: max |V| (one slot suffices for a single partial)
            *reinterpret_cast<float4*>(kvfin + ((size_t)seg * H + h) * KVP + 4 * e4) = tot;
            if (ksum_block) {
                if (el < DH / 4) *reinterpret_cast<float4*>(ksumT + ((size_t)tseg * H + h) * DH + 4 * el) = tot;   // ksum of the source
                if (el == 0) zsc[tseg * H + h] = zfold;
                // arrival counters of the target segment's fused InstanceNorm statistics (mlp.0 is enqueued behind this launch)
                if (h == 0 && el < STATCNT_PER_SEG) statcnt[tseg * STATCNT_PER_SEG + el] = 0;
            }
        }
    }
    if (ksum_block) return;
    __syncthreads();
    if (!op_thread) return;
    // M_t[mr + e][h*64 + 4 db + j] = sum_q W[mr + e][q] KV^T[4 db + j][q]:  four lanes per row (16 q each, fixed order), combined
    // by two exchange steps ((l0 + l1) + (l2 + l3) on every lane)
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (!(abl & 1)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 k4 = kvs[j * 16 + qq + 4 * i];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    acc[e][j] = fmaf(wv[e][i][0], k4.x, acc[e][j]);
                    acc[e][j] = fmaf(wv[e][i][1], k4.y, acc[e][j]);
                    acc[e][j] = fmaf(wv[e][i][2], k4.z, acc[e][j]);
                    acc[e][j] = fmaf(wv[e][i][3], k4.w, acc[e][j]);
                }
            }
    } else {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[e][j] = wv[e][j][0] + kvs[qq].x;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[e][j];
            const float o1 = __shfl_xor(v, 1);
            v = (qq & 1) ? o1 + v : v + o1;          // lower lane's value first on both lanes
            const float o2 = __shfl_xor(v, 2);
            acc[e][j] = (qq & 2) ? o2 + v : v + o2;
        }
    if (qq >= 2 || (abl & 2)) return;
    // lane qq = 0 stores row mr, lane qq = 1 row mr + 1 (all four lanes hold both sums)
    const int row = mr + qq;
    const float o0 = qq ? acc[1][0] : acc[0][0], o1 = qq ? acc[1][1] : acc[0][1], o2 = qq ? acc[1][2] : acc[0][2],
                o3 = qq ? acc[1][3] : acc[0][3];
    const int c = h * DH + 4 * db;   // first of this block's 4 columns of M_t
    if (prec == 0) {
        vf4 v = {o0, o1, o2, o3};
        // (write-through / non-temporal forms of this store -- sc1, sc0 sc1, nt -- were A/B-timed: no difference)
        *reinterpret_cast<vf4*>(mop_seg(Mop, tseg) + (size_t)row * MOP_LD + c) = v;
    } else {
        // split planes in the slab-major layout of the weight planes: (m, k) at ((k / 32) * 512 + m) * 32 + k % 32
        unsigned p0a, p1a, p2a = 0, p0b, p1b, p2b = 0;
        if (prec >= 3) {   // fp16 terms of the scaled operator
            fp16_split2(o0 * mscale, o1 * mscale, p0a, p1a);
            fp16_split2(o2 * mscale, o3 * mscale, p0b, p1b);
        } else {
            bf16_split3(o0, o1, p0a, p1a, p2a);
            bf16_split3(o2, o3, p0b, p1b, p2b);
        }
        unsigned short* pl = Mpl + (size_t)tseg * 3 * MPL_PLANE + ((size_t)(c >> 5) * 512 + row) * 32 + (c & 31);
        *reinterpret_cast<u32x2*>(pl) = (u32x2){p0a, p0b};
        *reinterpret_cast<u32x2*>(pl + MPL_PLANE) = (u32x2){p1a, p1b};
        if (prec == 2) *reinterpret_cast<u32x2*>(pl + 2 * MPL_PLANE) = (u32x2){p2a, p2b};
    }
}

// =====================================================================================================
// K4  mlp.0 with merge AND the linear-attention apply folded in:
//         u = W0a x + sum_h z_h (.) (M_h Qf_h) + (b0 + W0b bm)   [512 x N]
//     M_h = (W0b Wm)[:, head h] KV_h of the tile's segment (kv_final_kernel), Qf = elu(q) + 1 (qkv_kv_kernel),
//     z_h[n] = 1 / (Qf_h[:, n] . ksum_h + 1e-6)  -- K loop over [x ; Qf], AttnFoldHooks (gemm_f32_mfma.h)
//     (GATs_SuperGlue.py:78-79 linear attention, :101 merge, :113 cat, :122 first Conv1d) + per-tile InstanceNorm partials
//     (sum u, sum u^2 over the tile's real columns; :126).
// =====================================================================================================
// tiles (one InstanceNorm partial per 64-column tile whatever BN is)
using Mlp0TileW8 = GemmTile<128, MLP0_BN, 4, 2, false>;       // all arithmetics: 8 waves, one 32x32 MFMA tile each (fp32: 43.1 vs 45.0 us on 4 waves)
using Mlp0TileS = GemmTile<64, MLP0_BN, 2, 2, false, false, 2>;   // fp32, launches that leave CUs empty: 64x64, two K groups of 4 waves
// (round 2 also measured 128x128 tiles -- 16 waves fp32: kernel -3 %, frames/s in flight -0.8 %; 8 waves split-bf16: kernel -4 %,
//  frames/s equal; the attention fold needs thread = (k group, column) on a 64-column tile, they are gone)

// per-workgroup timeline of mlp0_kernel (tools/trace_mlp0.py): 8 x u64 per workgroup
// [hw_id, xcc_id, t_entry, shader cycles, t_after_mainloop, t_end, rt, ct], 100 MHz wall clock.
// The pointer is a kernel argument (nullptr = off); it can only be set in a -DGATSSPG_PROFILING_BUILD library.
#ifdef GATSSPG_PROFILING_BUILD
unsigned long long* g_trace = nullptr;
#else
static constexpr unsigned long long* g_trace = nullptr;
#endif

// ABL (profiling builds only, wrong results): main-loop ablations of gemm_mainloop_ex.  PREC as in qkv_kv_kernel.
// QF: quarter-fragment main loop (gemm_f32_mfma.h; fp32 arithmetic on the 8-wave tile only)
// SF: the fused InstanceNorm reducer (stat_last_block) is compiled in.  It is a tuning alternative (GATSSPG_STAT_FUSED=1), never taken in
//     the product -- but its 64 staging registers were what set the kernel's register count (118 of the 126), so the product
//     instantiations leave it out (round 6: found in the ISA of the 96-register build, whose only spills sat in that dead branch).
template <class T, int ABL = 0, int PREC = 0, int BT = 0, int QF = 0, int SF = 0>
__global__ __launch_bounds__(T::THREADS, (QF >= 2 ? 5 : PREC >= 2 ? 4 : 1)) void mlp0_kernel(const float* __restrict__ W0, const float* __restrict__ b0,
                                                   const unsigned short* __restrict__ Whi, const unsigned short* __restrict__ Wlo,
                                                   const unsigned short* __restrict__ Wl2,
                                                   const float* __restrict__ Z, const float* __restrict__ Qbuf,
                                                   const float* __restrict__ Mop, const unsigned short* __restrict__ Mpl,
                                                   const float* __restrict__ ksumT,
                                                   float* __restrict__ U, float* __restrict__ statpart, float* __restrict__ stats,
                                                   int* __restrict__ statcnt, ColLayout L, unsigned long long* trace) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (PREC >= 3) fp16_saturate_mode();
    const unsigned long long t_entry = trace ? wall_clock64() : 0;
    const unsigned long long c_entry = trace ? clock64() : 0;
    int rt, ct;
    constexpr int TPW = T::BN / MLP0_BN;   // 64-column tiles (= InstanceNorm partials) per workgroup
    constexpr int MT = 512 / T::BM;
    if (!xcd_tile_map(MT, active_tiles(L) / TPW, rt, ct)) return;
    ct = global_tile(L, ct * TPW) / TPW;   // windows and segments are multiples of 128 columns
    const int c0 = ct * T::BN, ld = L.ld;
    const float* A = W0 + (size_t)rt * T::BM * 512;
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % T::WAVES_MN;   // (K-split tiles: wave within its group)
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
    // requested before the main loop (see qkv_kv_kernel); the six-term and the fp16 loops have no 16 registers to park it in
    // (128-VGPR budget of two 8-wave workgroups per CU) and fetch it after the loop
    float bias[T::TM][16];
    auto load_bias = [&]() {
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const vf4 b4 = ldg4(b0 + rt * T::BM + (wm * T::TM + tm) * 32 + 8 * k + 4 * half);
                bias[tm][4 * k + 0] = b4[0]; bias[tm][4 * k + 1] = b4[1]; bias[tm][4 * k + 2] = b4[2]; bias[tm][4 * k + 3] = b4[3];
            }
    };
    float* btab = smem + smem_floats_mainloop<T, PREC>() + AttnFoldHooks::ZP_FLOATS;   // BT: bias through an LDS table (see qkv_kv_kernel)
    if constexpr (BT) {
        static_assert(T::BM == 128 && T::KS == 1, "half a piece of bias values, one wave group");
        if ((tid >> 6) == 0 && lane < 32) glds16(b0 + rt * T::BM + 4 * lane, btab);
    } else if constexpr (PREC < 2) {
        load_bias();
    }
    f32x16 acc[T::TM][T::TN];
    zero_acc(acc);
    // ABL == 5 (profiling): every workgroup streams the SAME weight panel and the SAME column tile (cache-hot operands)
    const float* Ah = ABL == 5 ? W0 : A;
    const int ch0 = ABL == 5 ? 0 : c0;
    const TileSeg ts = tile_seg(L, c0, T::BN);   // segments start on multiples of 128 columns: a tile never straddles two
    // K slabs 0..7: the x half of W0 against the state; slabs 8..15: this segment's message operator against Qf
    const float* Am = mop_seg(Mop, ts.seg) + (size_t)rt * T::BM * MOP_LD;
    static_assert(MOP_LD == 512, "the message operator shares the row stride of W0");
    auto al = [&](int kt) { return kt < 8 ? Ah + kt * BK : Am + (kt - 8) * BK; };
    auto bl = [&](int kt) { return (kt < 8 ? Z + (size_t)kt * BK * ld : Qbuf + (size_t)(kt - 8) * BK * ld) + ch0; };
    AttnFoldHooks hooks;
    hooks.init(ksumT + (size_t)ts.seg * H * DH, smem + smem_floats_mainloop<T, PREC>(), wn);
    if constexpr (PREC == 1 || PREC >= 3) {
        const size_t ro = (size_t)rt * T::BM * BK;   // slab-major planes (see qkv_kv_kernel); M_t planes in the same layout
        const unsigned short* Mh = Mpl + (size_t)ts.seg * 3 * MPL_PLANE + ro;
        auto ah = [&](int kt) { return kt < 8 ? Whi + ro + (size_t)kt * 512 * BK : Mh + (size_t)(kt - 8) * 512 * BK; };
        auto alo = [&](int kt) { return kt < 8 ? Wlo + ro + (size_t)kt * 512 * BK : Mh + MPL_PLANE + (size_t)(kt - 8) * 512 * BK; };
        gemm_mainloop_bf3<T, decltype(ah), decltype(alo), decltype(bl), AttnFoldHooks, (PREC >= 3), (PREC == 4 ? 4 : 3)>(
            acc, reinterpret_cast<unsigned short*>(smem), 512 / BK, ah, alo, BK, bl, ld, &hooks);
    } else if constexpr (PREC == 2) {
        const size_t ro = (size_t)rt * T::BM * BK;
        const unsigned short* Mh = Mpl + (size_t)ts.seg * 3 * MPL_PLANE + ro;
        gemm_mainloop_bf6<T>(
            acc, reinterpret_cast<unsigned short*>(smem), 512 / BK,
            [&](int kt, int pl) {
                return kt < 8 ? (pl == 0 ? Whi : pl == 1 ? Wlo : Wl2) + ro + (size_t)kt * 512 * BK
                              : Mh + (size_t)pl * MPL_PLANE + (size_t)(kt - 8) * 512 * BK;
            },
            BK, bl, ld, &hooks);
    } else {
        gemm_mainloop<T, decltype(al), decltype(bl), (ABL == 5 ? 0 : ABL), IdentityCol, AttnFoldHooks, QF>(acc, smem, 512 / BK, al, 512, bl, ld,
                                                                                                          IdentityCol(), &hooks);
    }
    acc[0][0] = hooks.kept;
    ksplit_reduce<T>(acc, smem);
    if constexpr (BT) read_bias16<T>(btab, wm, half, bias);
    else if constexpr (PREC >= 2) load_bias();
    const unsigned long long t_loop = trace ? wall_clock64() : 0;
    constexpr int TS = T::BN + 1;
    float* Tl = smem;  // [BM][BN + 1]
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * T::TM + tm) * 32 + mfma_row(r, half);
                const int col = (wn * T::TN + tn) * 32 + l31;
                Tl[row * TS + col] = acc[tm][tn][r] + bias[tm][r];
            }
    __syncthreads();
    {   // per-row (sum, centred sum of squares) of the real columns of each 64-column tile: THREADS / BM lanes per row,
        // each a fixed contiguous column range, combined by shuffles (fixed order).  One pass, shifted by the first
        // column of the row (a pivot within a few std of the mean), so M2 = sum d^2 - (sum d)^2 / n does not cancel even
        // when |mean| >> std; stat_final merges the tiles with Chan's formula.
        constexpr int LPR = T::THREADS / T::BM;    // lanes per row
        constexpr int LPS = LPR / TPW;             // lanes per (row, 64-column tile)
        constexpr int CPL = MLP0_BN / LPS;         // columns per lane
        static_assert(LPS >= 1, "at least one lane per row and 64-column tile");
        const int row = tid / LPR, q = tid % LPR, sub = q / LPS, part = q % LPS;
        const int valid = min(max(ts.valid - sub * MLP0_BN, 0), MLP0_BN);
        const float pivot = Tl[row * TS + sub * MLP0_BN];
        const float* tr = Tl + row * TS + sub * MLP0_BN + part * CPL;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < CPL; ++m) {
            const float t = tr[m];                                           // unconditional LDS read (a guarded one becomes a
            const float d = (part * CPL + m < valid) ? t - pivot : 0.f;      // branch + s_waitcnt per element), masked afterwards
            s1 += d;
            s2 += d * d;
        }
#pragma unroll
        for (int o = 1; o < LPS; o <<= 1) {
            s1 += __shfl_xor(s1, o);
            s2 += __shfl_xor(s2, o);
        }
        if (part == 0) {
            const float nv = (float)valid;
            const size_t t64 = (size_t)ct * TPW + sub;
            stat_partial_store(statpart + (t64 * 2 + 0) * 512 + rt * T::BM + row, nv * pivot + s1);                      // sum
            stat_partial_store(statpart + (t64 * 2 + 1) * 512 + rt * T::BM + row, nv > 0.f ? s2 - s1 * s1 / nv : 0.f);   // M2
        }
    }
    // the partial stores above go first; the tile's own stores follow them and may still be in flight when the ticket is drawn
    asm volatile("" ::: "memory");
    // then the tile leaves through LDS as 16-byte stores: 16 lanes cover one 256-byte row segment
#pragma unroll
    for (int idx = tid; idx < T::BM * (T::BN / 4); idx += T::THREADS) {
        const int row = idx / (T::BN / 4), c4 = (idx % (T::BN / 4)) * 4;
        const float* t = Tl + row * TS + c4;
        vf4 v = {t[0], t[1], t[2], t[3]};
        *reinterpret_cast<vf4*>(U + (size_t)(rt * T::BM + row) * ld + c0 + c4) = v;
    }
    constexpr int TILE_STORES = T::BM * (T::BN / 4) / T::THREADS;   // per thread, behind its partial stores
    static_assert(T::BM * (T::BN / 4) % T::THREADS == 0, "whole stores per thread");
    if constexpr (SF) {
        if (statcnt) stat_last_block<T, TILE_STORES>(statpart, stats, statcnt, L, ts, rt, smem);
    }
    if (trace && tid == 0) {
        unsigned long long* r = trace + (size_t)blockIdx.x * 8;
        r[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        r[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        r[2] = t_entry; r[3] = clock64() - c_entry; r[4] = t_loop; r[5] = wall_clock64(); r[6] = rt; r[7] = ct;
    }
}

// K5  InstanceNorm statistics per (segment, channel): mean and 1/sqrt(var + 1e-5), biased variance
//     (nn.InstanceNorm1d defaults, GATs_SuperGlue.py:126).  The per-tile partials are (sum_t, M2_t about the tile
//     mean); with n_t real columns per tile,  M2 = sum_t M2_t + sum_t sum_t^2 / n_t - S^2 / n  (Chan's merge written
//     out).  The cancellation-prone part is evaluated in double precision, where it is harmless, and the fp32 rounding
//     of sum_t enters only at second order (d M2 / d sum_t = 2 (mean_t - mean)).  Fixed order: 64 channels x 16
//     tile-ranges per block, ranges summed in tile order and combined in range order.
// TW: columns per partial -- MLP0_BN (64) from the channel-major mlp.0 kernels, 32 from mlp0_sp's transposed epilogue (one per wave strip).
// A block = ROWS channels x PARTS tile ranges; with twice the partials (TW = 32) it takes half the channels and twice the ranges, so that a
// range is still one batch of <= 8 loads at the headline shape (the first form kept 64 x 16 and paid a second dependent round trip: +1 us).
template <int TW>
__global__ __launch_bounds__(1024) void stat_final_kernel(const float* __restrict__ statpart, float* __restrict__ stats,
                                                          ColLayout L) {
    constexpr int ROWS = TW == 32 ? 32 : 64, PARTS = 1024 / ROWS;
    __shared__ double red[2][PARTS][ROWS];
    const int rl = threadIdx.x % ROWS, part = threadIdx.x / ROWS;
    const int seg = blockIdx.x, row = blockIdx.y * ROWS + rl;
    const int frame = seg >> 1, side = seg & 1;
    if (!((L.side_mask >> side) & 1)) return;
    const int t0 = (frame * L.np + (side ? L.n1p : 0)) / TW;
    const int nt = (side ? L.n2p : L.n1p) / TW;
    const int n = side ? L.n2 : L.n1;
    const int per = (nt + PARTS - 1) / PARTS;
    const int tb = part * per, te = min(nt, tb + per);
    double S = 0.0, QP = 0.0;
    for (int tt = tb; tt < te; tt += 8) {   // 2 x 8 loads in flight at a time on clamped addresses (see kv_final_kernel)
        float xs[8], xm[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t tile = (size_t)(t0 + min(tt + u, nt - 1));
            xs[u] = statpart[(tile * 2 + 0) * 512 + row];
            xm[u] = statpart[(tile * 2 + 1) * 512 + row];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = tt + u;
            const int nv = min(TW, n - t * TW);   // real columns of tile t of this segment (<= 0: pad-only tile)
            if (t < te && nv > 0) {
                const double st = (double)xs[u], mt = (double)xm[u];
                const double inv = nv == TW ? 1.0 / TW : 1.0 / nv;
                S += st;
                QP += mt + st * st * inv;
            }
        }
    }
    red[0][part][rl] = S;
    red[1][part][rl] = QP;
    __syncthreads();
    if (part == 0) {
        S = red[0][0][rl];
        QP = red[1][0][rl];
#pragma unroll
        for (int p = 1; p < PARTS; ++p) {
            S += red[0][p][rl];
            QP += red[1][p][rl];
        }
        const double mean = S / n;
        double var = (QP - S * mean) / n;
        if (var < 0.0) var = 0.0;
        stats[((size_t)seg * 2 + 0) * 512 + row] = (float)mean;
        stats[((size_t)seg * 2 + 1) * 512 + row] = (float)(1.0 / sqrt(var + 1e-5));
    }
}

// =====================================================================================================
// K6  mlp.3 with the InstanceNorm + ReLU applied on the B-operand load; the accumulators start from
//     residual + bias:  Z = (Z + b3) + W3 relu((u - mean) * rstd)      (GATs_SuperGlue.py:126-128, :59,64)
// =====================================================================================================
using Mlp3TileTallW8 = GemmTile<128, 64, 4, 2, false>;   // both arithmetics: 128x64 on 8 waves, 252 workgroups (fp32: 21.3 vs 24.6 us, 938 vs 914 frames/s one at a time)
using Mlp3Tile = GemmTile<64, 64, 2, 2, false>;          // alternative (tuning builds): 64x64 on 4 waves, 504 workgroups
using Mlp3TileS = GemmTile<64, 64, 2, 2, false, false, 2>;   // fp32, launches that leave CUs empty: 64x64, two K groups of 4 waves

// DS: the output tile leaves straight from the accumulators (store_tile_regs; plain 128 x 64 / 64 x 64 tiles, not the K-split one)
template <class T, int ABL = 0, int PREC = 0, int DS = 0, int QF = 0>
__global__ __launch_bounds__(T::THREADS, (PREC >= 2 ? 4 : 1)) void mlp3_kernel(const float* __restrict__ W3, const float* __restrict__ b3,
                                                   const unsigned short* __restrict__ Whi, const unsigned short* __restrict__ Wlo,
                                                   const unsigned short* __restrict__ Wl2,
                                                   const float* __restrict__ U, const float* __restrict__ stats,
                                                   float* __restrict__ Z, ColLayout L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (PREC >= 3) fp16_saturate_mode();
    int rt, ct;
    constexpr int MT = 256 / T::BM;
    constexpr int TPW = T::BN / 64;   // 64-column tiles per column tile of this kernel
    if (!xcd_tile_map(MT, active_tiles(L) / TPW, rt, ct)) return;
    ct = global_tile(L, ct * TPW) / TPW;
    const int c0 = ct * T::BN, ld = L.ld;
    const TileSeg ts = tile_seg(L, c0, T::BN);  // segments start on multiples of 128: a tile never straddles two
    const float* mean = stats + ((size_t)ts.seg * 2 + 0) * 512;
    const float* rstd = stats + ((size_t)ts.seg * 2 + 1) * 512;
    const float* A = W3 + (size_t)rt * T::BM * 512;
    f32x16 acc[T::TM][T::TN];
    // start from the residual + bias: the Z tile is fetched before the main loop instead of after it
    // (its latency hides under the GEMM; the epilogue becomes a pure store)
    {
        const int lane_ = threadIdx.x & 63, wave_ = (threadIdx.x >> 6) % T::WAVES_MN;
        const bool first_group = (int)(threadIdx.x >> 6) < T::WAVES_MN;   // K-split tiles: the second wave group starts from zero
        const int wm_ = wave_ / T::WN, wn_ = wave_ % T::WN, half_ = lane_ >> 5, l31_ = lane_ & 31;
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rt * T::BM + (wm_ * T::TM + tm) * 32 + mfma_row(r, half_);
                    const float v = Z[(size_t)row * ld + c0 + (wn_ * T::TN + tn) * 32 + l31_] + b3[row];
                    acc[tm][tn][r] = first_group ? v : 0.f;
                }
    }
    auto al = [&](int kt) { return A + kt * BK; };
    auto bl = [&](int kt) { return U + (size_t)kt * BK * ld + c0; };
    auto xm = [&](int kt) { return mean + kt * BK; };
    auto xr = [&](int kt) { return rstd + kt * BK; };
    if constexpr (PREC == 1 || PREC >= 3) {
        const size_t ro = (size_t)rt * T::BM * BK;   // slab-major planes (see qkv_kv_kernel)
        auto ah = [&](int kt) { return Whi + ro + (size_t)kt * 256 * BK; };
        auto alo = [&](int kt) { return Wlo + ro + (size_t)kt * 256 * BK; };
        auto bx1 = [](float v, float2 ms) { return fmaxf((v - ms.x) * ms.y, 0.f); };
        gemm_mainloop_bf3_ex<T, decltype(ah), decltype(alo), decltype(bl), decltype(xm), decltype(xr), decltype(bx1), true, NoHooks,
                             (PREC >= 3), (PREC == 4 ? 4 : 3)>(acc, reinterpret_cast<unsigned short*>(smem), 512 / BK, ah, alo, BK, bl, ld, xm,
                                                               xr, bx1);
    } else if constexpr (PREC == 2) {
        const size_t ro = (size_t)rt * T::BM * BK;
        auto ap = [&](int kt, int pl) { return (pl == 0 ? Whi : pl == 1 ? Wlo : Wl2) + ro + (size_t)kt * 256 * BK; };
        auto bx1 = [](float v, float2 ms) { return fmaxf((v - ms.x) * ms.y, 0.f); };
        gemm_mainloop_bf6_ex<T, decltype(ap), decltype(bl), decltype(xm), decltype(xr), decltype(bx1), true>(
            acc, reinterpret_cast<unsigned short*>(smem), 512 / BK, ap, BK, bl, ld, xm, xr, bx1);
    } else {
        auto bx = [](vf4& v, float2 ms) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = fmaxf((v[q] - ms.x) * ms.y, 0.f);
        };
        gemm_mainloop_ex<T, decltype(al), decltype(bl), decltype(xm), decltype(xr), decltype(bx), true, ABL, IdentityCol, NoHooks, QF>(
            acc, smem, 512 / BK, al, 512, bl, ld, xm, xr, bx);
    }
    ksplit_reduce<T>(acc, smem);
    if constexpr (DS && T::KS == 1) store_tile_regs<T>(acc, Z + (size_t)rt * T::BM * ld + c0, ld, [](int, float v) { return v; });
    else store_tile_via_lds<T>(acc, smem, Z + (size_t)rt * T::BM * ld + c0, ld, [](int, float v) { return v; });
}

// =====================================================================================================
// K7  final_proj + F.normalize(p=2, dim=channels, eps=1e-12)      (GATs_SuperGlue.py:209-213)
//     One workgroup owns all 256 output channels of a 32-column tile, so the L2 norm is an
//     in-block reduction.  Query-side tiles additionally leave point-major (MDT [b][n1p][256], 1 KiB rows): the score
//     GEMM then reads its A operand row-major like a weight matrix (b128 fragment reads) instead of [K][M].
// =====================================================================================================
using FinalTile = GemmTile<256, 32, 8, 1, false>;   // 8 waves x one 32x32 tile (4 waves x 64x32 left one wave per SIMD: 15.0 us)

// planes_prec: 0, or the arithmetic (2 bf16x6 / 4 fp16x4) whose 16-bit planes of the query descriptors (MDTp, slab-major over all b * n1p
// rows; fp16: of 2^SCORE_SPLIT_SCALE_LOG2 x) the split score contraction reads as its A operand
__global__ __launch_bounds__(FinalTile::THREADS) void final_proj_norm_kernel(const float* __restrict__ Wf, const float* __restrict__ bf,
                                                              const float* __restrict__ Z, float* __restrict__ MD,
                                                              float* __restrict__ MDT, unsigned short* __restrict__ MDTp,
                                                              int planes_prec, ColLayout L) {
    using T = FinalTile;
    if (planes_prec >= 3) fp16_saturate_mode();
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float npart[T::WM][32];
    const int ct = blockIdx.x;
    const int c0 = ct * T::BN, ld = L.ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    float bias[T::TM][16];   // requested before the main loop
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const vf4 b4 = ldg4(bf + (wave * T::TM + tm) * 32 + 8 * k + 4 * half);
            bias[tm][4 * k + 0] = b4[0]; bias[tm][4 * k + 1] = b4[1]; bias[tm][4 * k + 2] = b4[2]; bias[tm][4 * k + 3] = b4[3];
        }
    f32x16 acc[T::TM][T::TN];
    zero_acc(acc);
    gemm_mainloop<T>(
        acc, smem, D / BK, [&](int kt) { return Wf + kt * BK; }, D,
        [&](int kt) { return Z + (size_t)kt * BK * ld + c0; }, ld);
    float ss = 0.f;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[tm][0][r] + bias[tm][r];
            acc[tm][0][r] = v;
            ss += v * v;
        }
    ss += __shfl_xor(ss, 32);
    if (half == 0) npart[wave][l31] = ss;
    __syncthreads();
    float n2 = npart[0][l31];
#pragma unroll
    for (int w = 1; w < T::WM; ++w) n2 += npart[w][l31];   // fixed order
    const float nrm = sqrtf(n2);
    const float inv = 1.f / fmaxf(nrm, 1e-12f);
    const TileSeg ts = tile_seg(L, c0, T::BN);   // 32-column tiles never straddle a segment (segments are multiples of 128)
    constexpr int TS = D + 4;                    // point-major staging tile [32][260]
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wave * T::TM + tm) * 32 + mfma_row(r, half);
            const float v = acc[tm][0][r] * inv;
            MD[(size_t)row * ld + c0 + l31] = v;
            if (ts.side == 0) smem[l31 * TS + row] = v;   // block-uniform branch; the main loop ended on a barrier
        }
    if (ts.side == 0) {
        __syncthreads();
        float* dst = MDT + ((size_t)ts.frame * L.n1p + (c0 - ts.seg_start)) * D;
#pragma unroll
        for (int idx = tid; idx < 32 * (D / 4); idx += T::THREADS) {
            const int pt = idx / (D / 4), c4 = (idx % (D / 4)) * 4;
            *reinterpret_cast<vf4*>(dst + (size_t)pt * D + c4) = *reinterpret_cast<const vf4*>(smem + pt * TS + c4);
        }
        if (planes_prec) {
            // (point, slab, quarter) -> 8 consecutive channels -> one 16-byte piece of every plane; element (m, k) of a plane at
            // ((k / 32) * R + m) * 32 + k % 32, R = b * n1p rows
            const size_t R = (size_t)L.b * L.n1p;
            const size_t m0 = (size_t)ts.frame * L.n1p + (c0 - ts.seg_start);
            const float sc = planes_prec >= 3 ? (float)(1 << SCORE_SPLIT_SCALE_LOG2) : 1.f;
#pragma unroll
            for (int idx = tid; idx < 32 * 32; idx += T::THREADS) {
                const int pt = idx >> 5, slab = (idx >> 2) & 7, q = idx & 3;
                const float* v = smem + pt * TS + slab * 32 + q * 8;
                unsigned p0[4], p1[4], p2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (planes_prec >= 3) { fp16_split2(v[2 * e] * sc, v[2 * e + 1] * sc, p0[e], p1[e]); p2[e] = 0; }
                    else bf16_split3(v[2 * e], v[2 * e + 1], p0[e], p1[e], p2[e]);
                }
                unsigned short* d0 = MDTp + ((size_t)slab * R + m0 + pt) * 32 + q * 8;
                *reinterpret_cast<u32x4*>(d0) = (u32x4){p0[0], p0[1], p0[2], p0[3]};
                *reinterpret_cast<u32x4*>(d0 + R * D) = (u32x4){p1[0], p1[1], p1[2], p1[3]};
                if (planes_prec == 2) *reinterpret_cast<u32x4*>(d0 + 2 * R * D) = (u32x4){p2[0], p2[1], p2[2], p2[3]};
            }
        }
    }
}

// =====================================================================================================
// K8  score contraction + exp:  E[n][m] = exp( (sum_d A[n][d] B[d][m]) / scale_factor )
//     (GATs_SuperGlue.py:217 and the numerator of both softmaxes of :218; for |score| <= 80 the max-subtraction of
//     softmax is not needed for range).  E is written into the conf buffer; the tile's row sums (over its 64 columns)
//     and column sums (over its 128 rows) go to partial buffers that conf_finalize_kernel reduces in a fixed order.
//     RAW (1 / scale_factor > 80): the scaled scores themselves are written, no sums (max-subtracting path).
//     A = point-major query descriptors MDT (row-major [n][256]), B = channel-major 3D descriptors MD.
// =====================================================================================================
using ScoreTileW8 = GemmTile<SC_BM, SC_BN, 4, 2, false>;        // default: 128x64 on 8 waves: 880 tiles at 1000/7000 = 1.7 rounds of the 512 slots
using ScoreTileSq = GemmTile<SC_BM, 2 * SC_BN, 2, 4, false>;    // alternative (tuning builds): 128x128 on 8 waves, 440 tiles = one round
// Both one-round shapes measured SLOWER than 1.7 rounds of 128x64: 128x128 48.4 vs 45.3 us (event-timed), 256x64 (90 KB of LDS,
// one workgroup per CU) 52.7 vs 46.4 us; three 128x64 workgroups per CU (80 VGPRs) unchanged.

template <class T, bool RAW>
__global__ __launch_bounds__(T::THREADS) void score_exp_kernel(const float* __restrict__ MDT, const float* __restrict__ MD,
                                                               float* __restrict__ conf, float* __restrict__ rowpart,
                                                               float* __restrict__ colpart, ColLayout L, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nrt = (L.n1p + T::BM - 1) / T::BM, nct = L.n2p / T::BN;   // a last row tile may hang over n1p (rows masked below)
    int rt, ct;
    const int frame = blockIdx.y;
    if (!xcd_tile_map(nrt, nct, rt, ct)) return;
    const int ld = L.ld;
    const float* Ap = MDT + ((size_t)frame * L.n1p + rt * T::BM) * D;     // [M][K], row stride 256
    const float* Bp = MD + (size_t)frame * L.np + L.n1p + ct * T::BN;
    f32x16 acc[T::TM][T::TN];
    zero_acc(acc);
    gemm_mainloop<T>(
        acc, smem, D / BK, [&](int kt) { return Ap + kt * BK; }, D,
        [&](int kt) { return Bp + (size_t)kt * BK * ld; }, ld);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
    constexpr int TS = T::BN + 1;
    float* Tl = smem;  // [128][65]
    float* cf = conf + (size_t)frame * L.n1 * L.n2;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * T::TM + tm) * 32 + mfma_row(r, half);
            const int col = wn * 32 + l31;
            const int gi = rt * T::BM + row, gj = ct * T::BN + col;
            const float sc = acc[tm][0][r] / scale;
            Tl[row * TS + col] = (gi < L.n1 && gj < L.n2) ? (RAW ? sc : expf(sc)) : 0.f;
        }
    __syncthreads();
    // the tile leaves through LDS: 16 lanes cover one 256-byte row segment (16-byte stores when the rows of conf are
    // 16-byte aligned, i.e. n2 % 4 == 0 and an aligned base; otherwise 4-byte stores, 64 lanes per row segment)
    if ((L.n2 & 3) == 0 && (reinterpret_cast<uintptr_t>(cf) & 15) == 0) {
        for (int idx = tid; idx < T::BM * (T::BN / 4); idx += T::THREADS) {
            const int row = idx / (T::BN / 4), c4 = (idx % (T::BN / 4)) * 4;
            const int gi = rt * T::BM + row, gj = ct * T::BN + c4;
            if (gi < L.n1 && gj < L.n2) {
                const float* t = Tl + row * TS + c4;
                vf4 v = {t[0], t[1], t[2], t[3]};
                *reinterpret_cast<vf4*>(cf + (size_t)gi * L.n2 + gj) = v;
            }
        }
    } else {
        for (int idx = tid; idx < T::BM * T::BN; idx += T::THREADS) {
            const int row = idx / T::BN, col = idx % T::BN;
            const int gi = rt * T::BM + row, gj = ct * T::BN + col;
            if (gi < L.n1 && gj < L.n2) cf[(size_t)gi * L.n2 + gj] = Tl[row * TS + col];
        }
    }
    if constexpr (!RAW) {
        // row sums: THREADS / BM lanes per row; column sums: one wave per THREADS / 64-th of the rows (conflict-free
        // column walks); fixed order throughout
        constexpr int LPR = T::THREADS / T::BM, CPL = T::BN / LPR;
        const int row = tid / LPR, hp = tid % LPR;
        const float* tr = Tl + row * TS + hp * CPL;
        float s = 0.f;
#pragma unroll 8
        for (int m = 0; m < CPL; ++m) s += tr[m];
#pragma unroll
        for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o);
        if (hp == 0 && rt * T::BM + row < L.n1p) rowpart[((size_t)frame * nct + ct) * L.n1p + rt * T::BM + row] = s;
        constexpr int NQ = T::THREADS / T::BN, RPQ = T::BM / NQ;   // NQ row groups of RPQ rows, one thread per (group, column)
        const int c = tid % T::BN, qp = tid / T::BN;
        float t = 0.f;
#pragma unroll 8
        for (int m = 0; m < RPQ; ++m) t += Tl[(qp * RPQ + m) * TS + c];
        __syncthreads();
        Tl[qp * T::BN + c] = t;   // re-use the tile head for the NQ x BN part sums
        __syncthreads();
        if (tid < T::BN) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q) tot += Tl[q * T::BN + tid];
            colpart[((size_t)frame * nrt + rt) * L.n2p + ct * T::BN + tid] = tot;
        }
    }
}

// =====================================================================================================
// Generic dense GEMM used off the hot path (GATs with_linear_transform=True:  out = elu(W^T pre (+h)) ).
//   C[256][cols of Y segments] = elu( sum_c W[c][o] * P[c][n]  (+ R[o][n]) )       (GATs.py:57,62,65,70)
// =====================================================================================================
using WltTile = GemmTile<64, 64, 2, 2, true>;

__global__ __launch_bounds__(256) void gats_wlt_kernel(const float* __restrict__ W, const float* __restrict__ P,
                                                       float* __restrict__ Z, ColLayout L, int add_h) {
    using T = WltTile;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int rt, ct;
    const int NT = L.ld / T::BN;
    if (!xcd_tile_map(4, NT, rt, ct)) return;
    const int c0 = ct * T::BN, ld = L.ld;
    const TileSeg ts = tile_seg(L, c0, T::BN);
    if (ts.side == 0) return;  // only the 3D side is touched by a GATs layer
    f32x16 acc[T::TM][T::TN];
    zero_acc(acc);
    gemm_mainloop<T>(
        acc, smem, D / BK, [&](int kt) { return W + (size_t)kt * BK * D + rt * 64; }, D,
        [&](int kt) { return P + (size_t)kt * BK * ld + c0; }, ld);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / T::WN, wn = wave % T::WN, half = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = rt * 64 + wm * 32 + mfma_row(r, half);
        float* p = Z + (size_t)row * ld + c0 + wn * 32 + l31;
        float v = acc[0][0][r];
        if (add_h) v += *p;
        *p = elu1(v);
    }
}

// Power-of-two scales of the fp16 planes (AttnW::SC): one block per (matrix, layer); s = 2^(13 - floor(log2 max|w|)) puts the
// largest entry in [2^13, 2^14) -- a factor 4 below the fp16 maximum -- so that the second fp16 term of an entry stays a normal
// number down to |w| ~ 2^-24 max|w| (unscaled, every weight below 2^-3 had a subnormal second term: 2^-25 absolute error).
__global__ __launch_bounds__(1024) void weight_scale_kernel(float* __restrict__ packed) {
    __shared__ float red[1024];
    const int m = blockIdx.x, layer = blockIdx.y;
    float* blk = packed + PW_ATTN + (size_t)layer * AttnW::SIZE;
    float mx = 0.f;
    if (m == 0) {
        for (int e = threadIdx.x; e < 768 * 256; e += 1024) mx = fmaxf(mx, fabsf(blk[AttnW::WQKV + e]));
    } else if (m == 1) {   // the x half of mlp.0 only: the message half reaches the matrix pipe through the operator planes of kv_final
        for (int e = threadIdx.x; e < 512 * 256; e += 1024) mx = fmaxf(mx, fabsf(blk[AttnW::W0 + (size_t)(e >> 8) * 512 + (e & 255)]));
    } else {
        for (int e = threadIdx.x; e < 256 * 512; e += 1024) mx = fmaxf(mx, fabsf(blk[AttnW::W3 + e]));
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float top = red[0];
        int e = 0;
        if (top > 0.f && top < 3.0e38f) e = min(max(13 - ilogbf(top), -40), 60);
        blk[AttnW::SC + m] = __builtin_ldexpf(1.f, e);
        if (m == 0) blk[AttnW::SC + 3] = 1.f;
    }
    if (m == 1) {   // row-L1 norms of the message half per head (AttnW::SC[4 + h]): |M_h[r][d]| <= l1_h * max |KV_h| (kv_final_kernel)
        for (int h = 0; h < H; ++h) {
            __syncthreads();
            float l1 = 0.f;
            if (threadIdx.x < 512) {
                const float* wr = blk + AttnW::W0 + (size_t)threadIdx.x * 512 + 256 + h * DH;
                for (int q = 0; q < DH; ++q) l1 += fabsf(wr[q]);
            }
            red[threadIdx.x] = l1;
            __syncthreads();
            for (int o = 512; o > 0; o >>= 1) {
                if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
                __syncthreads();
            }
            if (threadIdx.x == 0) blk[AttnW::SC + 4 + h] = red[0];
        }
    }
}

// one-time split of the three big operators of every attention layer into bf16 hi / lo planes (AttnWB layout)
__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ packed, unsigned short* __restrict__ packedb) {
    fp16_saturate_mode();
    const int layer = blockIdx.y;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const float* src = packed + PW_ATTN + (size_t)layer * AttnW::SIZE;
    unsigned short* dst = packedb + (size_t)layer * AttnWB::SIZE;
    constexpr size_t NQ = 768 * 256, N0 = 512 * 512, N3 = 256 * 512;
    // planes are stored SLAB-MAJOR: element (m, k) of an [M][K] operator at ((k / 32) * M + m) * 32 + k % 32, so that the
    // 128-row x 32-k slab a workgroup stages per step is 8 KB of consecutive bytes (full 128-byte lines per load instruction
    // instead of 64-byte row pieces 2 * K bytes apart)
    float x, sx;   // sx: the matrix's fp16 scale (weight_scale_kernel)
    size_t hi, lo, lo2, d, h16, l16;
    auto slab_major = [](size_t i, size_t M, size_t K) { const size_t m = i / K, k = i % K; return ((k >> 5) * M + m) * 32 + (k & 31); };
    if (e < NQ) { x = src[AttnW::WQKV + e]; sx = src[AttnW::SC + 0]; d = slab_major(e, 768, 256); hi = AttnWB::QKV_HI + d; lo = AttnWB::QKV_LO + d; lo2 = AttnWB::QKV_LO2 + d; h16 = AttnWB::QKV_H16 + d; l16 = AttnWB::QKV_L16 + d; }
    else if (e < NQ + N0) { x = src[AttnW::W0 + (e - NQ)]; sx = src[AttnW::SC + 1]; d = slab_major(e - NQ, 512, 512); hi = AttnWB::W0_HI + d; lo = AttnWB::W0_LO + d; lo2 = AttnWB::W0_LO2 + d; h16 = AttnWB::W0_H16 + d; l16 = AttnWB::W0_L16 + d; }
    else if (e < NQ + N0 + N3) { x = src[AttnW::W3 + (e - NQ - N0)]; sx = src[AttnW::SC + 2]; d = slab_major(e - NQ - N0, 256, 512); hi = AttnWB::W3_HI + d; lo = AttnWB::W3_LO + d; lo2 = AttnWB::W3_LO2 + d; h16 = AttnWB::W3_H16 + d; l16 = AttnWB::W3_L16 + d; }
    else return;
    const unsigned h = bf16_rne_bits(x);
    const float r1 = x - __uint_as_float(h << 16);
    const unsigned m = bf16_rne_bits(r1);
    dst[hi] = (unsigned short)h;
    dst[lo] = (unsigned short)m;
    dst[lo2] = (unsigned short)bf16_rne_bits(r1 - __uint_as_float(m << 16));
    unsigned fh, fl;   // fp16 terms: the conversion of the main loop (fp16_split2), so weights and activations split alike
    fp16_split2(x * sx, 0.f, fh, fl);
    dst[h16] = (unsigned short)(fh & 0xFFFFu);
    dst[l16] = (unsigned short)(fl & 0xFFFFu);
}

void launch_split_weights(float* packed, unsigned short* packedb, hipStream_t s) {
    constexpr size_t N = 768 * 256 + 512 * 512 + 256 * 512;
    hipLaunchKernelGGL(weight_scale_kernel, dim3(3, 8), dim3(1024), 0, s, packed);
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((N + 255) / 256), 8), dim3(256), 0, s, packed, packedb);
}

// ------------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------------

// Kernels whose dynamic LDS request exceeds the 64 KiB default need the limit raised once per device.  The once-flag
// lives in a function template instantiated per KERNEL (the kernel is a non-type template argument), so two variants
// that merely share a signature never share it.
template <auto Kernel>
void allow_big_lds() {
    static bool done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !done[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024 - 2048);
        if (dev >= 0 && dev < 64) done[dev] = true;
    }
}

// fp32 kernels (8-wave tiles): output tiles straight from the accumulators.  bit 0: the Q tiles of qkv_kv, bit 1: mlp3.  Tuning builds read
// GATSSPG_FP32_DIRECT per launch (tools/ab_live.py); the product takes the default.
constexpr int FP32_DIRECT_DEFAULT = 3;
[[maybe_unused]] static int fp32_direct() { return tuning_knob("FP32_DIRECT", FP32_DIRECT_DEFAULT); }

template <class T, int PREC, int BT = 0, int DS = 0, int QF = 0>
static void launch_qkv_t(const float* Wqkv, const float* bqkv, const unsigned short* wb, const Workspace& w, hipStream_t s,
                         ProfileHook* hk) {
    const int NT = active_tiles(w.L);
    allow_big_lds<qkv_kv_kernel<T, PREC, BT, DS, QF>>();
    GATSSPG_LAUNCH(hk, KID_QKV_KV, s, (qkv_kv_kernel<T, PREC, BT, DS, QF>), dim3(xcd_grid(6, NT)), dim3(T::THREADS), (smem_bytes<T, PREC>() + 512 * BT), s,
                   Wqkv, bqkv, wb ? wb + (PREC >= 3 ? AttnWB::QKV_H16 : AttnWB::QKV_HI) : nullptr,
                   wb ? wb + (PREC >= 3 ? AttnWB::QKV_L16 : AttnWB::QKV_LO) : nullptr, wb ? wb + AttnWB::QKV_LO2 : nullptr, w.Z,
                   w.Q, w.kvpart, w.L);
}

// fp16 modes run on the LDS-DMA loop only (their planes carry the pack-time scale that only those kernels undo).  The bf16 modes stay
// on the first form: A/B-timed on one box (profiles/r04_split_loop_ab.txt), mlp0 bf16x3 25.0 (first form) vs 25.2 us, 1866 vs 1842
// frames/s in flight; bf16x6 34.4 vs 40.9 us, 1397 vs 1285 -- the three-plane stage makes the DMA loop 1.5x the LDS traffic.  Tuning
// builds switch them with GATSSPG_SPLIT_LOOP_BF16X3 / _BF16X6 = 1.
bool stat_fused() {
    // measured (profiles/r04_stat_fused_ab.txt, one box, three alternations): the fused form makes mlp0 5.3 us longer (partial-store
    // acknowledgement -> ticket atomic -> agent-scope loads of the partials: three dependent round trips through the memory side for
    // the last workgroup) and saves the 4.8 us launch: 1.001 vs 0.983 ms per frame at the headline shape, 1231 vs 1232 frames/s in
    // flight; +1.7 % in flight at 500 x 2000.  The separate launch stays the default; tuning builds: GATSSPG_STAT_FUSED=1.
    return tuning_knob("STAT_FUSED", 0) != 0;   // (read per launch: tools/ab_live.py)
}

bool split_loop_glds(int prec) {
    const int b3 = tuning_knob("SPLIT_LOOP_BF16X3", 0), b6 = tuning_knob("SPLIT_LOOP_BF16X6", 0);   // (read per launch: tools/ab_live.py)
    return prec >= 3 || (prec == 1 && b3 != 0) || (prec == 2 && b6 != 0);
}

// fp32 arithmetic on the LDS-DMA loop (tuning builds: GATSSPG_FP32_DMA, bit 0 qkv_kv, 1 mlp0, 2 mlp3); shapes whose launches leave CUs
// empty keep the K-split tiles of the register-staged loop
// fp32 qkv_kv / mlp0 (8-wave tiles): bias through an LDS table filled by half an LDS-DMA piece at kernel entry (read_bias16) instead of per-lane loads
// (tuning builds; measured -0.2 % in flight / -0.7 % one at a time at the headline shape: the split loop's gain from its table came with 38 fewer
//  registers per wave, which this loop does not get)
[[maybe_unused]] static int fp32_bias_table() { return tuning_knob("FP32_BIAS_TABLE", 0); }
// register diet of the fp32 mlp0 kernel (tuning builds, read per launch): 0 = the two-half fragment loop, 1 = quarter fragments,
// 2 = quarter fragments + bias through the LDS table
// Product default (-1 = by shape): launches of more than DIET_MIN_TILES 64-column tiles take the dieted kernels -- quarter fragments + bias
// table: mlp0 94 VGPRs (from 126), qkv_kv 80 (from 98), zero scratch, bit-identical results -- smaller launches the two-half loop.
// Interleaved A/Bs in one process (profiles/r06b_ab_live_register_diet_*.txt, r06c_*): headline in flight +0.5 % / +0.8 % (two boxes), one at
// a time -0.2 ... +0.1 %; 8 frames per step +0.8 % both ways; 500 x 2000 -1 ... -2 % in flight (hence the threshold).  mlp3's quarter-fragment
// form (76 VGPRs) is 2 % slower one frame at a time: tuning builds only.
constexpr int DIET_MIN_TILES = 64;
[[maybe_unused]] static int mlp0_diet() { return tuning_knob("MLP0_DIET", -1); }
[[maybe_unused]] static int qkv_diet() { return tuning_knob("QKV_DIET", -1); }
[[maybe_unused]] static int mlp3_diet() { return tuning_knob("MLP3_DIET", 0); }
static int fp32_dma(const Workspace& w) {
    const int m = tuning_knob("FP32_DMA", 0);
    return (w.prec == 0 && active_tiles(w.L) > 64) ? m : 0;
}

void launch_qkv_kv(const float* Wqkv, const float* bqkv, const unsigned short* wb, const Workspace& w, hipStream_t s,
                   ProfileHook* hk) {
    if (fp32_dma(w) & 1) return launch_qkv_kv_dma(Wqkv, bqkv, w, s, hk);
    if (split_loop_glds(w.prec)) {
        // Wqkv is the first member of the layer's AttnW block: its scales sit at AttnW::SC from there
        launch_qkv_kv_sp(Wqkv - AttnW::WQKV + AttnW::SC, bqkv, wb, w, s, hk);
        return;
    }
    static const int tq = tuning_knob("QKV_BTILE", 0);   // tuning builds: 1 = split-bf16 on the 4-wave tile
    if (w.prec == 1 && tq == 1) launch_qkv_t<QkvTileB, 1>(Wqkv, bqkv, wb, w, s, hk);
    else if (w.prec == 1) launch_qkv_t<QkvTileW8, 1>(Wqkv, bqkv, wb, w, s, hk);
    else if (w.prec == 2) launch_qkv_t<QkvTileW8, 2>(Wqkv, bqkv, wb, w, s, hk);
#ifdef GATSSPG_TUNING
    else if (qkv_diet() == 1) launch_qkv_t<QkvTileW8, 0, 0, (FP32_DIRECT_DEFAULT & 1), 1>(Wqkv, bqkv, wb, w, s, hk);   // quarter fragments
    else if (qkv_diet() == 0 && fp32_bias_table()) launch_qkv_t<QkvTileW8, 0, 1>(Wqkv, bqkv, wb, w, s, hk);
    else if (qkv_diet() == 0 && !(fp32_direct() & 1)) launch_qkv_t<QkvTileW8, 0>(Wqkv, bqkv, wb, w, s, hk);
    else if (qkv_diet() == 0) launch_qkv_t<QkvTileW8, 0, 0, (FP32_DIRECT_DEFAULT & 1)>(Wqkv, bqkv, wb, w, s, hk);
    else if (qkv_diet() == 2 || active_tiles(w.L) > DIET_MIN_TILES)
        launch_qkv_t<QkvTileW8, 0, 1, (FP32_DIRECT_DEFAULT & 1), 1>(Wqkv, bqkv, wb, w, s, hk);   // quarter fragments + bias table (80 VGPRs)
#else
    else if (active_tiles(w.L) > DIET_MIN_TILES) launch_qkv_t<QkvTileW8, 0, 1, (FP32_DIRECT_DEFAULT & 1), 1>(Wqkv, bqkv, wb, w, s, hk);
#endif
#ifdef GATSSPG_TUNING
    else if (!(fp32_direct() & 1)) launch_qkv_t<QkvTileW8, 0>(Wqkv, bqkv, wb, w, s, hk);
#endif
    else launch_qkv_t<QkvTileW8, 0, 0, (FP32_DIRECT_DEFAULT & 1)>(Wqkv, bqkv, wb, w, s, hk);   // small launches: the two-half fragment loop
}

void launch_kv_final(const float* W0, const Workspace& w, int cross, const float* kv_src, hipStream_t s, ProfileHook* hk) {
    static const int abl = tuning_knob("KVF_ABL", 0);   // tuning builds: timing-only ablations of the operator phase
    // one workgroup per d block by default since round 4: every partial read once (interleaved A/B, profiles/r04_ab_live_kv_final.txt: kernel
    // 11.4 -> 10.5 us event-timed, +0.7 ... +1.6 % frames/s in flight at the three shapes, bit-identical results); KVF_RS=2: the two-row-half form
    // (round 6, second experiment: the projection SPLIT in two launches -- first the four [K_h ; V_h] row tiles (504 workgroups: one round of
    //  the 512 resident slots), then ONE grid holding the two Q row tiles and kv_final's workgroups in the narrow 512-thread form -- so that the
    //  reduction + message operator runs beside the Q tiles instead of in front of mlp.0.  Bit-identical.  Event-timed at the headline shape:
    //  27.5 + 21.6 us against 39.2 + 11.8 us for the classic pair, frame 0.9948 vs 0.9942 ms, 1264 vs 1266 frames/s in flight; 500 x 2000:
    //  0.626 vs 0.567 ms.  One Q tile per CU is a single workgroup's dependent chain (17 us for 8 slabs), not a matrix-pipe load: what the
    //  hidden kv_final saves the lonely Q tiles give back.  With kv_final's workgroups FIRST in the grid they took both slots of half the
    //  CUs and the Q tiles paired up on the rest: 1.046 ms.  profiles/r06e_*, r06f_*; removed.)
    // (round 6: a narrow form -- two d rows per workgroup, 264 workgroups of 512 threads so that every CU pulls partials -- was built, verified
    //  bit-identical and A/B-timed: 11.85 vs 11.22 us event-timed, 994.8 vs 1001.1 frames/s one at a time; the reduction is not bound by
    //  what one CU can have in flight.  profiles/r06c_ab_live_kvf_narrow_*.txt; removed)
    if (tuning_knob("KVF_RS", 1) == 1) {
        GATSSPG_LAUNCH(hk, KID_KV_FINAL, s, kv_final_kernel<1>, dim3(17, w.nseg * H), dim3(1024), 0, s, w.kvpart, kv_src, w.kvfin, W0, w.Mop, w.Mpl,
                       w.ksumT, w.zsc, w.statcnt, W0 - AttnW::W0 + AttnW::SC, w.L, cross, w.prec, abl);
        return;
    }
    GATSSPG_LAUNCH(hk, KID_KV_FINAL, s, kv_final_kernel<2>, dim3(33, w.nseg * H), dim3(1024), 0, s, w.kvpart, kv_src, w.kvfin,
                   W0, w.Mop, w.Mpl, w.ksumT, w.zsc, w.statcnt, W0 - AttnW::W0 + AttnW::SC, w.L, cross, w.prec, abl);
}

template <class T, int ABL, int PREC, int BT = 0, int QF = 0, int SF = 0>
static void launch_mlp0_body(const float* W0, const float* b0, const unsigned short* wb, const Workspace& w, hipStream_t s,
                             ProfileHook* hk);
template <class T, int ABL, int PREC, int BT = 0, int QF = 0>
static void launch_mlp0_t(const float* W0, const float* b0, const unsigned short* wb, const Workspace& w, hipStream_t s,
                          ProfileHook* hk) {
    // fp32: the reducer is compiled in only where it can run (tuning builds with GATSSPG_STAT_FUSED=1).  The split-bf16 instantiations keep it
    // (never taken: statcnt stays nullptr): their register allocation sits at the 128 cap, and without the dead branch the six-term kernel
    // picked up a 12-byte spill -- they stay exactly the round-5 kernels.
    if constexpr (PREC != 0) {
        launch_mlp0_body<T, ABL, PREC, BT, QF, 1>(W0, b0, wb, w, s, hk);
    } else {
#ifdef GATSSPG_TUNING
        if (stat_fused()) return launch_mlp0_body<T, ABL, PREC, BT, QF, 1>(W0, b0, wb, w, s, hk);
#endif
        launch_mlp0_body<T, ABL, PREC, BT, QF, 0>(W0, b0, wb, w, s, hk);
    }
}
template <class T, int ABL, int PREC, int BT, int QF, int SF>
static void launch_mlp0_body(const float* W0, const float* b0, const unsigned short* wb, const Workspace& w, hipStream_t s,
                             ProfileHook* hk) {
    allow_big_lds<mlp0_kernel<T, ABL, PREC, BT, QF, SF>>();
    const int NT = active_tiles(w.L) / (T::BN / MLP0_BN);
    GATSSPG_LAUNCH(hk, KID_MLP0, s, (mlp0_kernel<T, ABL, PREC, BT, QF, SF>), dim3(xcd_grid(512 / T::BM, NT)), dim3(T::THREADS),
                   (smem_bytes<T, PREC>() + sizeof(float) * AttnFoldHooks::ZP_FLOATS + 512 * BT), s, W0, b0,
                   wb ? wb + (PREC >= 3 ? AttnWB::W0_H16 : AttnWB::W0_HI) : nullptr, wb ? wb + (PREC >= 3 ? AttnWB::W0_L16 : AttnWB::W0_LO) : nullptr,
                   wb ? wb + AttnWB::W0_LO2 : nullptr, w.Z, w.Q, w.Mop, w.Mpl, w.ksumT, w.U,
                   w.statpart, w.stats, (SF && stat_fused()) ? w.statcnt : nullptr, w.L, g_trace);
}
template <class T, int ABL, int PREC, int DS = 0, int QF = 0>
static void launch_mlp3_t(const float* W3, const float* b3, const unsigned short* wb, const Workspace& w, hipStream_t s,
                          ProfileHook* hk) {
    allow_big_lds<mlp3_kernel<T, ABL, PREC, DS, QF>>();
    const int NT = active_tiles(w.L) / (T::BN / 64);
    GATSSPG_LAUNCH(hk, KID_MLP3, s, (mlp3_kernel<T, ABL, PREC, DS, QF>), dim3(xcd_grid(256 / T::BM, NT)), dim3(T::THREADS),
                   (smem_bytes<T, PREC>()), s, W3, b3, wb ? wb + (PREC >= 3 ? AttnWB::W3_H16 : AttnWB::W3_HI) : nullptr,
                   wb ? wb + (PREC >= 3 ? AttnWB::W3_L16 : AttnWB::W3_LO) : nullptr, wb ? wb + AttnWB::W3_LO2 : nullptr, w.U,
                   w.stats, w.Z, w.L);
}

void launch_mlp(const float* W0, const float* b0, const float* W3, const float* b3, const unsigned short* wb, const Workspace& w,
                hipStream_t s, ProfileHook* hk) {
    // MLP0_TILE / MLP3_TILE / MLP0_BTILE select the alternative (equally correct) tile shapes in tuning builds; the ablation
    // variants (wrong results, timing only) exist only in a -DGATSSPG_PROFILING_BUILD library.
    static const int t0 = tuning_knob("MLP0_TILE", 0), t3 = tuning_knob("MLP3_TILE", 1);
    // Launches that leave CUs empty (few columns: OnePose's own 500 x 2000 operating point) are bound by ONE workgroup's
    // dependent MFMA chain, not by the matrix pipes: they take the 64x64 tile with the K loop split over two wave groups
    // (4x / 2x the workgroups, half the chain per wave).  SMALL_NT = largest number of 64-column tiles that still does.
    static const int small_nt0 = tuning_knob("SMALL_NT0", 0), small_nt3 = tuning_knob("SMALL_NT3", 64);
    const bool small0 = w.prec == 0 && active_tiles(w.L) <= small_nt0, small3 = w.prec == 0 && active_tiles(w.L) <= small_nt3;
    (void)t0;
    const bool sp = split_loop_glds(w.prec);
    const float* sc = W0 - AttnW::W0 + AttnW::SC;
    const int dma = fp32_dma(w);
    if (dma & 2) launch_mlp0_dma(W0, b0, w, s, hk);
    else if (sp) launch_mlp0_sp(sc, b0, wb, w, s, hk);
    else if (small0 && t0 == 0) launch_mlp0_t<Mlp0TileS, 0, 0>(W0, b0, wb, w, s, hk);
    else if (w.prec == 1) launch_mlp0_t<Mlp0TileW8, 0, 1>(W0, b0, wb, w, s, hk);
    else if (w.prec == 2) launch_mlp0_t<Mlp0TileW8, 0, 2>(W0, b0, wb, w, s, hk);
#ifdef GATSSPG_PROFILING_BUILD
    else if (t0 == 11) launch_mlp0_t<Mlp0TileW8, 1, 0>(W0, b0, wb, w, s, hk);   // no global loads in the loop
    else if (t0 == 12) launch_mlp0_t<Mlp0TileW8, 2, 0>(W0, b0, wb, w, s, hk);   // no loads, no LDS writes
    else if (t0 == 15) launch_mlp0_t<Mlp0TileW8, 5, 0>(W0, b0, wb, w, s, hk);   // all workgroups stream the same (cache-hot) panels
    else if (t0 == 16) launch_mlp0_t<Mlp0TileW8, 6, 0>(W0, b0, wb, w, s, hk);   // every load L1-hot
#endif
#ifdef GATSSPG_TUNING
    else if (mlp0_diet() == 1) launch_mlp0_t<Mlp0TileW8, 0, 0, 0, 1>(W0, b0, wb, w, s, hk);   // quarter fragments, bias in registers (110 VGPRs)
    else if (mlp0_diet() == 0 && fp32_bias_table()) launch_mlp0_t<Mlp0TileW8, 0, 0, 1>(W0, b0, wb, w, s, hk);
    else if (mlp0_diet() == 0) launch_mlp0_t<Mlp0TileW8, 0, 0>(W0, b0, wb, w, s, hk);
    else if (mlp0_diet() == 2 || active_tiles(w.L) > DIET_MIN_TILES) launch_mlp0_t<Mlp0TileW8, 0, 0, 1, 1>(W0, b0, wb, w, s, hk);
#else
    else if (active_tiles(w.L) > DIET_MIN_TILES) launch_mlp0_t<Mlp0TileW8, 0, 0, 1, 1>(W0, b0, wb, w, s, hk);   // quarter fragments + bias table (94 VGPRs)
#endif
    else launch_mlp0_t<Mlp0TileW8, 0, 0>(W0, b0, wb, w, s, hk);   // small launches: the two-half fragment loop (126 VGPRs)
    // the InstanceNorm reducer is a launch of its own (stat_final_kernel: measured faster one frame at a time than finishing the statistics
    // inside the mlp.0 launch by its last workgroups -- stat_last_block; that form is GATSSPG_STAT_FUSED=1 in tuning builds)
    if (sp && !(dma & 2) && sp_ut_on(w.prec))
        GATSSPG_LAUNCH(hk, KID_STAT_FINAL, s, stat_final_kernel<32>, dim3(w.nseg, 16), dim3(1024), 0, s, w.statpart, w.stats, w.L);
    else if (!stat_fused()) GATSSPG_LAUNCH(hk, KID_STAT_FINAL, s, stat_final_kernel<MLP0_BN>, dim3(w.nseg, 8), dim3(1024), 0, s, w.statpart, w.stats, w.L);
    if (dma & 4) launch_mlp3_dma(W3, b3, w, s, hk);
    else if (sp) launch_mlp3_sp(sc, b3, wb, w, s, hk);
    else if (small3 && t3 == 1) launch_mlp3_t<Mlp3TileS, 0, 0>(W3, b3, wb, w, s, hk);
    else if (w.prec == 1 && t3 == 0) launch_mlp3_t<Mlp3Tile, 0, 1>(W3, b3, wb, w, s, hk);
    else if (w.prec == 1) launch_mlp3_t<Mlp3TileTallW8, 0, 1>(W3, b3, wb, w, s, hk);
    else if (w.prec == 2) launch_mlp3_t<Mlp3TileTallW8, 0, 2>(W3, b3, wb, w, s, hk);
#ifdef GATSSPG_PROFILING_BUILD
    else if (t3 == 13) launch_mlp3_t<Mlp3Tile, 3, 0>(W3, b3, wb, w, s, hk);   // steady-state loop cut: fixed cost only
#endif
#ifdef GATSSPG_TUNING
    else if (t3 == 1 && mlp3_diet() == 1) launch_mlp3_t<Mlp3TileTallW8, 0, 0, ((FP32_DIRECT_DEFAULT >> 1) & 1), 1>(W3, b3, wb, w, s, hk);   // quarter fragments
    else if (t3 == 1 && !(fp32_direct() & 2)) launch_mlp3_t<Mlp3TileTallW8, 0, 0>(W3, b3, wb, w, s, hk);
#endif
    else if (t3 == 1) launch_mlp3_t<Mlp3TileTallW8, 0, 0, ((FP32_DIRECT_DEFAULT >> 1) & 1)>(W3, b3, wb, w, s, hk);
    else launch_mlp3_t<Mlp3Tile, 0, 0>(W3, b3, wb, w, s, hk);
}

void launch_final_proj_norm(const float* Wf, const float* bf, const Workspace& w, hipStream_t s, ProfileHook* hk) {
    allow_big_lds<final_proj_norm_kernel>();
    GATSSPG_LAUNCH(hk, KID_FINAL_PROJ, s, final_proj_norm_kernel, dim3(w.L.ld / FinalTile::BN), dim3(FinalTile::THREADS),
                   (smem_bytes<FinalTile>()), s, Wf, bf, w.Z, w.MD, w.MDT, w.MDTp, score_on_split_loop(w.prec, 0) ? w.prec : 0, w.L);
}

static bool score_square() {
    static const int t = tuning_knob("SCORE_TILE", 0);   // 0 (default): 128x64; 1 (tuning builds): 128x128
    return t == 1;
}
int score_tile_rows() { return SC_BM; }
int score_tile_cols() { return score_square() ? ScoreTileSq::BN : ScoreTileW8::BN; }

template <class T, bool RAW>
static void launch_score_t(const Workspace& w, float* conf, float scale, hipStream_t s, ProfileHook* hk) {
    allow_big_lds<score_exp_kernel<T, RAW>>();
    const int nrt = (w.L.n1p + T::BM - 1) / T::BM;
    GATSSPG_LAUNCH(hk, KID_SCORE_EXP, s, (score_exp_kernel<T, RAW>), dim3(xcd_grid(nrt, w.L.n2p / T::BN), w.L.b), dim3(T::THREADS),
                   (smem_bytes<T>()), s, w.MDT, w.MD, conf, w.rowpart, w.colpart, w.L, scale);
}

static bool score_square();
// The fp32-class split modes (bf16x6: operands split exactly; fp16x4: the exact product of 22-bit operands) also run the score
// contraction on the 16-bit pipe; the three-term modes keep the fp32 MFMA here (their 2^-16 / dropped-term error would sit directly on
// the logits of the dual softmax).  The max-subtracting path (tiny scale factors) stays fp32 as well.
bool score_on_split_loop(int prec, int shifted) {
    const int on = tuning_knob("SCORE_SPLIT", 1);   // (read per launch: tools/ab_live.py)
    return on != 0 && !shifted && (prec == 2 || prec == 4) && !score_square();   // (its partial sums are per 64-column tile)
}

void launch_score_exp(const Workspace& w, float* conf, float scale, int shifted, hipStream_t s, ProfileHook* hk) {
    if (score_on_split_loop(w.prec, shifted)) return launch_score_exp_sp(w, conf, scale, s, hk);
    const bool sq = score_square();
    if (shifted && sq) launch_score_t<ScoreTileSq, true>(w, conf, scale, s, hk);
    else if (shifted) launch_score_t<ScoreTileW8, true>(w, conf, scale, s, hk);
    else if (sq) launch_score_t<ScoreTileSq, false>(w, conf, scale, s, hk);
    else launch_score_t<ScoreTileW8, false>(w, conf, scale, s, hk);
}

void launch_gats_wlt(const float* W, const float* P, const Workspace& w, int add_h, hipStream_t s, ProfileHook* hk) {
    const int NT = w.L.ld / WltTile::BN;
    GATSSPG_LAUNCH(hk, KID_GATS_WLT, s, gats_wlt_kernel, dim3(xcd_grid(4, NT)), dim3(256), (smem_bytes<WltTile>()), s, W, P, w.Z,
                   w.L, add_h);
}

}  // namespace gatsspg
