// K10: the adaptive model of BWTC levels 6..9 on the GPU.
//
// Replaces the model half of lib/BWTC.js:105-133: FenwickModel.encode / _rescale / _sumTree
// (lib/FenwickModel.js:47-87, 137-172).  The model's state lives per block (BWTC.js:105 builds a new one for every
// block), and what it hands to the range coder for a symbol - (sy_f, lt_f, tot_f) - depends on the block's symbols
// only, not on the coder.  So every block of a batch runs its model here, ONE WAVE per block, and the host is left
// with RangeCoder.encodeFreq (lib/RangeCoder.js:79-89) over the finished triples: one division per symbol instead of
// a tree walk with up to ten read-modify-writes (and a second one for the escape of a first occurrence).
//
// The tree (2 * numSyms <= 520 packed u32: high half = frequency, low half = escape count) sits in LDS.  One symbol
// at a time (the model is a serial recurrence), but the ~10 levels of its leaf-to-root path are lanes: lane l owns node
// (leaf >> l): reads it, reads its left sibling when the node is a right child (lt_f is the sum of those), adds the
// update.  The siblings are not on the path, so reading all levels at once equals the reference's bottom-up walk.
// Rescaling (every ~127 symbols) halves the leaves 64 at a time and rebuilds the inner nodes level by level.
// All u32 wrap-around arithmetic on the packed words is kept exactly as the reference's `>>> 0` arithmetic.
#include "pipeline.h"

#define K10_MAXSYM 260
#define K10_OVERFLOW 0xFFFFFFFFu

__device__ __forceinline__ void k10_sum_tree(u32* tree, u32 numSyms, u32 lane) {      // lib/FenwickModel.js:167-172
    // tree[i] = tree[2i] + tree[2i+1] for i = numSyms-1 .. 1: children have a longer bit length, so go level by level
    for (int bl = 9; bl >= 1; bl--) {
        const u32 lo = 1u << (bl - 1), hi = (1u << bl) < numSyms ? (1u << bl) : numSyms;
        for (u32 i = lo + lane; i < hi; i += 64u) tree[i] = tree[2u * i] + tree[2u * i + 1u];
        __builtin_amdgcn_wave_barrier();
    }
}

// out: sylt[b][k] = sy_f | lt_f << 16, tot[b][k] = tot_f for the k-th encodeFreq call of block b; ntri[b] = their number.
// A block emits one triple per symbol PLUS one per escape (a first occurrence, or a symbol whose frequency was scaled down to 0:
// ~1.3 % of the symbols on random bytes, up to ~18 % on adversarial input), so the rows hold `ocap` = 2 x stride triples (the
// round-2 rows of `stride` overflowed into the next block on incompressible input); a block that would not fit even there
// stops, reports K10_OVERFLOW and is modelled on the host (bwtc_block).
__global__ __launch_bounds__(64) void k10_model(const u16* A, u32 stride, const u32* pos, const u32* alpha, u32* sylt, u32* tot,
                                                u32* ntri, u32 ostride, u32 ocap) {
    __shared__ u32 tree[2 * K10_MAXSYM + 8];
    const u32 b = blockIdx.x, lane = threadIdx.x;
    const u32 nsym = pos[b] ? pos[b] - 1u : 0u;               // K2 appended bzip2's end-of-block symbol; BWTC has none
    const u32 size = alpha[b] + 1u;                           // FenwickModel(coder, alphabetSize + 1, ...)  lib/BWTC.js:105
    const u32 numSyms = size + 1u, increment = 0x0100u, max_prob = 0xFF00u;
    const u16* sym = A + (size_t)b * stride;
    u32* osl = sylt + (size_t)b * ostride;
    u32* oto = tot + (size_t)b * ostride;
    // init :13-32
    for (u32 i = lane; i < 2u * numSyms; i += 64u) tree[i] = i >= numSyms ? (i == numSyms + size ? increment << 16 : 1u) : 0u;
    __builtin_amdgcn_wave_barrier();
    k10_sum_tree(tree, numSyms, lane);
    u32 nout = 0;                                             // triples emitted (uniform)
    u32 my_sl = 0, my_to = 0;                                 // lane (nout & 63) holds the pending triple of its row
    // one call of FenwickModel.encode's body for leaf `symbol` in the domain chosen by (mask, shift), adding `update`
    auto walk = [&](u32 symbol, u32 mask, u32 shift, u32 update) {
        const u32 leaf = numSyms + symbol;
        const u32 L = 32u - (u32)__clz((int)leaf);            // nodes on the path incl. the root
        const u32 node = leaf >> lane;
        const bool on = lane < L;
        const u32 val = on ? tree[node] : 0u;
        u32 sib = (on && node > 1u && (node & 1u)) ? tree[node - 1u] : 0u;
        for (u32 off = 8; off > 0; off >>= 1) sib += __shfl_xor(sib, (int)off);      // L <= 10 lanes: sum over 16
        const u32 lt = (u32)__builtin_amdgcn_readlane((int)sib, 0);
        const u32 sy = (u32)__builtin_amdgcn_readlane((int)val, 0);
        const u32 to = (u32)__builtin_amdgcn_readlane((int)val, (int)(L - 1u));
        __builtin_amdgcn_wave_barrier();
        if (on) tree[node] = val + update;
        __builtin_amdgcn_wave_barrier();
        const u32 sl = ((sy & mask) >> shift) | (((lt & mask) >> shift) << 16);
        if (lane == (nout & 63u)) { my_sl = sl; my_to = (to & mask) >> shift; }
        nout++;
        if ((nout & 63u) == 0u) { osl[nout - 64u + lane] = my_sl; oto[nout - 64u + lane] = my_to; }
    };
    // _rescale :137-166 (wave-uniform call)
    auto rescale = [&]() {
        bool noEscape = true;
        for (u32 j = lane; j < numSyms - 1u; j += 64u) {
            u32 p = tree[numSyms + j];
            if (p & 0xFFFFu) { noEscape = false; continue; }
            p = (p & 0xFFFEFFFEu) >> 1;
            if (p == 0u) { p = 1u; noEscape = false; }
            tree[numSyms + j] = p;
        }
        const bool allNo = __ballot(!noEscape) == 0ull;
        if (lane == 0) {
            u32 p = tree[2u * numSyms - 1u];
            p = (p & 0xFFFEFFFEu) >> 1;
            if (allNo) p = 0u; else if (p == 0u) p = 1u << 16;
            tree[2u * numSyms - 1u] = p;
        }
        __builtin_amdgcn_wave_barrier();
        k10_sum_tree(tree, numSyms, lane);
    };
    // The wave runs alone on its SIMD (a launch has one wave per block and far fewer blocks than SIMDs): every instruction costs
    // 5-9 clocks, an LDS round trip ~50, a v_readlane into an SGPR and its use 20-30 (tests/microbench/lone_wave.hip).  So the
    // common path is kept short: no exec masking (lanes above the path read and rewrite tree[0], which is 0 and stays 0: node
    // = leaf >> lane is 0 there, and so is the "left sibling" index of the root), outputs go into their row by v_writelane,
    // and the next row of symbols is requested a row ahead.
    const u32 sh = lane < 31u ? lane : 31u;                   // leaf < 1024: lanes 10.. see node 0
    u32 nxt = lane < nsym ? sym[lane] : 0u;
    for (u32 r0 = 0; r0 < nsym; r0 += 64u) {
        if (nout + 192u > ocap) {                             // a row adds at most 128 triples (wave-uniform)
            if (lane == 0) ntri[b] = K10_OVERFLOW;
            return;
        }
        const u32 mine = nxt;
        nxt = r0 + 64u + lane < nsym ? sym[r0 + 64u + lane] : 0u;
        const u32 rows = nsym - r0 < 64u ? nsym - r0 : 64u;
        for (u32 t = 0; t < rows; t++) {
            const u32 s = (u32)__builtin_amdgcn_readlane((int)mine, (int)t);
            // path nodes and left siblings of all levels in ONE LDS round trip, the sibling sum over the row of 16 lanes by DPP
            // row_shr adds, leaf / root values by v_readlane, the new root from registers
            const u32 leaf = numSyms + s;
            const u32 L = 32u - (u32)__clz((int)leaf);
            const u32 node = leaf >> sh;
            const u32 val = tree[node];
            u32 sib = tree[(node & 1u) ? node - 1u : 0u];     // left sibling of a right child; the root's "sibling" is tree[0] = 0
            pin_vgpr(sib);                                    // both reads in flight before the branch below waits for the first
            const u32 leafv = (u32)__builtin_amdgcn_readlane((int)val, 0);
            if ((leafv & 0xFFFF0000u) == 0u) {                // never seen (or scaled away): the escape symbol first  :53-57
                const u32 esc = numSyms - 1u;
                const u32 ev = tree[numSyms + esc];
                u32 eupd = increment << 16;
                if ((tree[1] & 0xFFFFu) == 1u) eupd = 0u - ev;                        // the last escape: zero it out  :58-60
                walk(esc, 0xFFFF0000u, 16u, eupd);
                if (((tree[1] & 0xFFFF0000u) >> 16) >= max_prob) rescale();           // :85 inside the escape's own encode
                walk(s, 0x0000FFFFu, 0u, (increment << 16) - 1u);
                if (((tree[1] & 0xFFFF0000u) >> 16) >= max_prob) rescale();
                continue;
            }
            sib += (u32)__builtin_amdgcn_update_dpp(0, (int)sib, 0x111, 0xf, 0xf, true);          // row_shr:1
            sib += (u32)__builtin_amdgcn_update_dpp(0, (int)sib, 0x112, 0xf, 0xf, true);          // row_shr:2
            sib += (u32)__builtin_amdgcn_update_dpp(0, (int)sib, 0x114, 0xf, 0xf, true);          // row_shr:4
            sib += (u32)__builtin_amdgcn_update_dpp(0, (int)sib, 0x118, 0xf, 0xf, true);          // row_shr:8: lane 15 = sum of lanes 0..15
            const u32 lt = (u32)__builtin_amdgcn_readlane((int)sib, 15);
            const u32 to = (u32)__builtin_amdgcn_readlane((int)val, (int)(L - 1u));
            const u32 upd = increment << 16;                  // (the escape symbol itself never occurs in the data)
            tree[node] = val + (lane < L ? upd : 0u);
            __builtin_amdgcn_wave_barrier();
            my_sl = (u32)cjs_writelane((int)((leafv >> 16) | ((lt >> 16) << 16)), (int)(nout & 63u), (int)my_sl);
            my_to = (u32)cjs_writelane((int)(to >> 16), (int)(nout & 63u), (int)my_to);
            nout++;
            if ((nout & 63u) == 0u) { osl[nout - 64u + lane] = my_sl; oto[nout - 64u + lane] = my_to; }
            if (((to + upd) >> 16) >= max_prob) rescale();
        }
    }
    if (nout & 63u) { if (lane < (nout & 63u)) { osl[(nout & ~63u) + lane] = my_sl; oto[(nout & ~63u) + lane] = my_to; } }
    if (lane == 0) ntri[b] = nout;
}

int k10_model_run(Pipe P, u32* sylt, u32* tot, u32* ntri, u32 ostride, u32 ocap, hipStream_t stream) {
    hipLaunchKernelGGL(k10_model, dim3(P.g.nb), dim3(64), 0, stream, (const u16*)P.A, P.g.stride, (const u32*)P.pos, (const u32*)P.alpha, sylt, tot, ntri, ostride, ocap);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
