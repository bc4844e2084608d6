// K10: the adaptive model of BWTC levels 6..9 on the GPU.
//
// Replaces the model half of lib/BWTC.js:105-133: FenwickModel.encode / _rescale / _sumTree
// (lib/FenwickModel.js:47-87, 137-172).  The model's state lives per block (BWTC.js:105 builds a new one for every
// block), and what it hands to the range coder for a symbol - (sy_f, lt_f, tot_f) - depends on the block's symbols
// only, not on the coder.  So every block of a batch runs its model here, ONE WAVE per block, and the host is left
// with RangeCoder.encodeFreq (lib/RangeCoder.js:79-89) over the finished triples: one division per symbol instead of
// a tree walk with up to ten read-modify-writes (and a second one for the escape of a first occurrence).
//
// The tree (2 * numSyms <= 520 packed u32: high half = frequency, low half = escape count) sits in LDS.
// Round 4: the model is a serial recurrence symbol by symbol only where it has to be.  Between two rescales every symbol
// that has been seen before just adds `increment` to the nodes of its path, so for a CHUNK of up to 64 consecutive seen
// symbols with no rescale inside (the chunk ends where the root reaches max_prob: that symbol is its last) the triples are
// closed forms of the tree at the chunk's start: with c_eq / c_lt = the earlier symbols of the chunk that are equal to /
// to the left (in the tree's in-order) of symbol x of lane l,
//     sy = leaf[x] + inc * c_eq,   lt = (sum of the left siblings of x's path) + inc * c_lt,   tot = root + inc * l
// (every leaf to the left of x lies under exactly one left sibling of x's path; all sums on the packed words, mod 2^32, as the
// reference's `>>> 0` arithmetic - node = sum of its leaves either way).  A lane per symbol: its own ten-level walk, the two
// counts from a 64-step v_readlane loop; then the leaves get their increments (LDS atomics) and the inner nodes are re-summed.
// A symbol that has not been seen (or was scaled away) ends the chunk before it and takes the serial path - the escape
// symbol, then the symbol in the escape domain, each a walk with the ~10 levels of the path as lanes - as in round 3, where
// EVERY symbol took such a walk (~800 clocks per symbol for a wave alone on its SIMD: 133 ms per 900 kB block, all of it
// latency in front of the host's range coder).  Rescaling (every ~127 symbols) halves the leaves 64 at a time and rebuilds
// the inner nodes level by level.
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
        if (lane == 0) { osl[nout] = sl; oto[nout] = (to & mask) >> shift; }
        nout++;
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
    // 5-9 clocks, an LDS round trip ~50, a v_readlane into an SGPR and its use 20-30 (tests/microbench/lone_wave.hip).
    const u32 inc = increment << 16;
    u32 posn = 0;
    // the symbols of rows wbase / wbase + 64 (aligned rows of 64) in two registers; a chunk starts anywhere in the first and takes
    // its symbols by two shuffles - the next row is requested as soon as the window slides, a chunk or two before it is needed
    u32 wbase = 0;
    u32 r0 = lane < nsym ? sym[lane] : 0u, r1 = 64u + lane < nsym ? sym[64u + lane] : 0u;
    auto window = [&]() -> u32 {
        while (posn - wbase >= 64u) { wbase += 64u; r0 = r1; r1 = wbase + 64u + lane < nsym ? sym[wbase + 64u + lane] : 0u; }
        const u32 off = posn - wbase + lane;                  // 0 .. 126
        const u32 a0 = (u32)__shfl((int)r0, (int)(off & 63u)), a1 = (u32)__shfl((int)r1, (int)(off & 63u));
        return off < 64u ? a0 : a1;
    };
    u32 nxt = window();
    while (posn < nsym) {
        if (nout + 192u > ocap) {                             // a chunk adds at most 64 triples, an unseen symbol two (wave-uniform)
            if (lane == 0) ntri[b] = K10_OVERFLOW;
            return;
        }
        const u32 x = nxt;
        const u32 avail = nsym - posn < 64u ? nsym - posn : 64u;
        const u32 root = tree[1];
        // symbols until the root reaches max_prob (the one that gets it there included): root_hi + 0x100 * k >= 0xFF00
        const u32 root_hi = root >> 16;
        const u32 kmax = root_hi >= max_prob ? 1u : (max_prob - root_hi + increment - 1u) / increment;
        const u32 leafw = lane < avail ? tree[numSyms + x] : inc;          // (idle lanes count as seen)
        const u64 unseen = __ballot((leafw & 0xFFFF0000u) == 0u);
        u32 m = unseen ? (u32)__ffsll((long long)unseen) - 1u : 64u;      // seen symbols in front of the first unseen one
        if (m > avail) m = avail;
        if (m > kmax) m = kmax;
        if (m == 0u) {
            // the symbol at posn has never been seen (or was scaled away): the escape symbol first  :53-57, one symbol, serial
            const u32 sx = (u32)__builtin_amdgcn_readlane((int)x, 0);
            const u32 esc = numSyms - 1u;
            const u32 ev = tree[numSyms + esc];
            u32 eupd = inc;
            if ((tree[1] & 0xFFFFu) == 1u) eupd = 0u - ev;                            // the last escape: zero it out  :58-60
            walk(esc, 0xFFFF0000u, 16u, eupd);
            if (((tree[1] & 0xFFFF0000u) >> 16) >= max_prob) rescale();               // :85 inside the escape's own encode
            walk(sx, 0x0000FFFFu, 0u, inc - 1u);
            if (((tree[1] & 0xFFFF0000u) >> 16) >= max_prob) rescale();
            posn += 1u;
            nxt = window();
            continue;
        }
        const bool on = lane < m;
        // left siblings of the lane's own path (packed sum), leaf and root at the chunk's start
        u32 lt = 0;
        {
            u32 node = numSyms + x;
#pragma unroll
            for (int lv = 0; lv < 10; lv++) {
                const bool take = on && node > 1u && (node & 1u);
                const u32 sv = tree[take ? node - 1u : 0u];                           // tree[0] = 0
                lt += take ? sv : 0u;
                node >>= 1;
            }
        }
        // earlier symbols of the chunk equal to mine / to the LEFT of mine in the tree: the leaves under the left siblings of a path
        // are the leaves before it in the tree's in-order, and with numSyms no power of two the leaves sit on two levels - in-order is
        // the order of the leaf indices shifted to a common bit length, not the order of the symbols
        const u32 leafi = numSyms + x;
        const u32 ko = leafi << (10u - (32u - (u32)__clz((int)leafi)));             // (leaf < 1024: bit length <= 10)
        // both counts from ten ballots, most significant bit first: the lanes that agree with me on the bits so far and have a 0 where
        // I have a 1 are to my left (a 63-step v_readlane loop did the same in ~3 000 clocks of the wave's ~20 000 per chunk)
        u32 ceq, clt = 0;
        {
            const u64 lt_lanes = lanemask_lt();
            u64 same = __ballot(on) & lt_lanes;               // earlier lanes of the chunk that still agree with my key
#pragma unroll
            for (int bit = 9; bit >= 0; bit--) {
                const bool mine1 = (ko >> bit) & 1u;
                const u64 ones = __ballot(((ko >> bit) & 1u) != 0u);
                clt += mine1 ? (u32)__popcll(same & ~ones) : 0u;
                same &= mine1 ? ones : ~ones;
            }
            ceq = (u32)__popcll(same);
        }
        const u32 sy = leafw + inc * ceq, ltv = lt + inc * clt, to = root + inc * lane;
        if (on) {
            osl[nout + lane] = (sy >> 16) | ((ltv >> 16) << 16);
            oto[nout + lane] = to >> 16;
            atomicAdd(&tree[numSyms + x], inc);
        }
        nout += m;
        posn += m;
        nxt = window();
        __builtin_amdgcn_wave_barrier();
        k10_sum_tree(tree, numSyms, lane);
        if ((tree[1] >> 16) >= max_prob) rescale();
    }
    if (lane == 0) ntri[b] = nout;
}
int k10_model_run(Pipe P, u32* sylt, u32* tot, u32* ntri, u32 ostride, u32 ocap, hipStream_t stream) {
    hipLaunchKernelGGL(k10_model, dim3(P.g.nb), dim3(64), 0, stream, (const u16*)P.A, P.g.stride, (const u32*)P.pos, (const u32*)P.alpha, sylt, tot, ntri, ostride, ocap);
    HIP_CHECK_RET(hipGetLastError());
    return CJS_OK;
}
