// K1: batched cyclic suffix sort + BWT (replaces BWT.bwtransform2, lib/BWT.js:372-417).
#pragma once
#include "cjs_common.h"

#define K1_DEEP_SUB 64u     // list sub-regions per XCD region of the lane kernels (power of two)
#define K1_STAT_BIGROT 116     // stats[116..123]: rotations in 8-byte groups of more than 64 members, counted by k1f_bsort (8 spread words)
#define K1_STAT_FRONT_BIG 104  // stats[104]: buckets of the sample-sort front end that did not fit LDS (105..110: K1F_TRACE stage clocks)
#define K1_STAT_PUREROT 124    // stats[124]: rotations in front-end buckets of ONE 8-byte key (> 64 members) or beyond LDS, counted by k1f_scan
#define K1_DEEP_LANE 8u        // groups up to this size go to the lane kernels (k1_deep_pairs / k1_deep_small): one lane each
#define K1R_MAXR 40            // refinement rounds at most
#ifndef K1_RCS
#define K1_RCS 32u              // words between two blocks' list counters (rcnt): one 128-byte line each - the counters of the blocks the 8 XCDs work on at the
#endif                          // same time shared a line, and device-scope atomics on one line are serialised in the fabric (round 5)
// per-block counters of the doubling rounds (dcnt, dchg, dtot, dred): block b's word sits at K1_BI(B, b) of its row - the blocks one XCD works on
// (b mod 8) share lines, blocks of different XCDs never do (rstride = 8 groups of >= 32 words)
#define K1_BI(B, b) (((b) & 7u) * ((B).rstride >> 3) + ((b) >> 3))
#define K1_RCNT(B, round, b) ((B).rcnt[((size_t)(round) * (B).rnb8 + (b)) * K1_RCS])
#ifndef K1F_LEVELS
#define K1F_LEVELS 6u
#endif
// K1F_LEVELS: task levels of the front end (partition, then sort: 16 bytes of depth per sorting level since round 5; what is left when they run out stays a group
// for the doubling rounds).  An empty level is a launch on every sub-batch's critical path: 14 -> 6 levels = 7.48 -> 7.35 ms per 10^8-byte enwik step (median of 16)
#define K1_STAT_RTRACE 128      // stats[128..135]: K1F_TRACE builds, stage clocks of k1r_round
#define K1_MED_MAX 4096     // doubling rounds: largest group a workgroup sorts in LDS
#define K1_STATS 144
// Round 6: with 16-byte keys in the bucket sort every group the front end leaves is at least 16 bytes deep - also on the path
// without text stages (HTML-like input), where a bucket of ONE 8-byte key beyond LDS used to stay a single 8-deep group and kept
// the doubling rounds at h = 8: such a bucket is now partitioned by its next 8 bytes (one task level) and sorted 16 bytes deeper in
// LDS, and the rounds start at h = 16 (the round with h = 8 was the most expensive one: 1.9 of 10.8 ms per E8S-A step).  Whoever
// leaves a shallower group after all (the last task level on adversarial input) says so here, and the host starts the rounds at 8.
#ifndef K1_DEEP_START
#define K1_DEEP_START 1
#endif
// Round 6: a key that fills m >= 3 quantiles of the sample used to get ONE bucket [v, v + 1) and leave m - 2 EMPTY ones behind it (the equal splitters); on
// HTML-like input such buckets hold a quarter of the rotations and most are beyond LDS (E8S-A: 22.8 M rotations in 9 900 buckets of 2 300 on average - the
// task levels' work, 1.0 ms per step).  k1f_hist now spreads a heavy key's rotations over its own bucket AND the empty ones behind it by the bucket of their
// NEXT 8 bytes - against sub-splitters that k1f_sample takes from the key's OWN samples (the next 8 bytes of the sampled rotations that start with the key,
// sorted: quantiles of the conditional distribution; the global buckets of the next 8 bytes were tried first and split nothing - what follows `<td clas` is
// nearly always the same bucket) - monotone in bytes 8..16: the order between the sub-buckets is the order of the rotations, and k1f_bsort sorts them like
// any other bucket.
#ifndef K1F_SUBBUCKETS
#define K1F_SUBBUCKETS 1
#endif
#define K1D_SHALLOW(B) ((B).dbn[(K1D_MAXR + 1u) * 4u])      // (a word of dbn's last row, which no round uses: read back together with dtot)
// list entries of the refinement rounds (k1r_round) and the doubling rounds (k1_dbl.hip): one per rotation that still ties,
// a group = consecutive entries:  (group length - 1) << 52 | index in the group << 44 | rotation index << 22 | suffix-array position
#define K1E_POS(e) ((u32)(e) & 0x3FFFFFu)
#define K1E_S(e) ((u32)((e) >> 22) & 0x3FFFFFu)
#define K1E_IDX(e) ((u32)((e) >> 44) & 0xFFu)
#define K1E_LEN(e) (((u32)((e) >> 52) & 0xFFu) + 1u)
#define K1E_KEEP (1ull << 60)          // doubling rounds: the rotation's rank (= position of its group head) did not change this round
#define K1E_MAKE(len1, idx, s, pos) (((u64)(len1) << 52) | ((u64)(idx) << 44) | ((u64)(s) << 22) | (u64)(pos))
// Round 5: the byte in FRONT of a rotation - what the BWT gather (k1_finish) fetches with one random load per rotation, 10^8 of them per step - travels with the
// rotation through the text stages instead.  k1f_scatter packs it into the index word it writes (bits 24..31 over a 22-bit index); the bucket sorts and the task
// levels move packed words and write U next to the suffix array; the refinement rounds' entries carry it in the CARRY layout below (blocks below 2^20 bytes: 20-bit
// fields), the lane kernels permute U with the suffix array.  Blocks whose order is finished by anything else (doubling rounds, periodic routes) are gathered as before.
#define K1_SMASK 0x3FFFFFu
#define K1_SPACK(s, byte) ((u32)(s) | ((u32)(byte) << 24))
#define K1C_POS(e) ((u32)(e) & 0xFFFFFu)
#define K1C_S(e) ((u32)((e) >> 20) & 0xFFFFFu)
#define K1C_BYTE(e) ((u32)((e) >> 40) & 0xFFu)
#define K1C_IDX(e) ((u32)((e) >> 48) & 0xFFu)
#define K1C_LEN(e) (((u32)((e) >> 56) & 0xFFu) + 1u)
#define K1C_MAKE(len1, idx, byte, s, pos) (((u64)(len1) << 56) | ((u64)(idx) << 48) | ((u64)(byte) << 40) | ((u64)(s) << 20) | (u64)(pos))
#define K1_CARRY_MAXN (1u << 20)
// doubling rounds (k1_dbl.hip)
#define K1D_MAXR 24            // rounds at most (h0 >= 8, n < 2^22: 19 doublings + the tie-break round)
#ifndef K1D_GS
#define K1D_GS 256u
#endif
// K1D_GS: groups up to this size are list entries (ranked by counting inside a tile); larger ones are descriptors
#define K1_DM_SUB 64u      // sub-lists per class of the medium rounds (one counter each: a single counter serialises millions of appends)
#define K1_SPREAD 128
#define K1_MED_MAX 4096     // sparse phase: largest group a workgroup sorts in LDS

// sample-sort front end (k1_front.hip)
#ifndef K1F_NB
#define K1F_NB 2048         // buckets per block (power of two; k1f_bsort / k1f_hist ms per 10^8 rotations: 512 buckets x 4096-rotation slots 8.5 / 0.36,
                            // 1024 x 2048 3.40 / 0.49, 2048 x 1024 2.72 / 0.64: smaller LDS footprint = more workgroups per CU for a latency-bound kernel)
#endif
#define K1F_LOG_NB (K1F_NB == 512 ? 9u : K1F_NB == 1024 ? 10u : K1F_NB == 2048 ? 11u : K1F_NB == 4096 ? 12u : 0u)
static_assert(K1F_LOG_NB != 0u, "K1F_NB: 512, 1024, 2048 or 4096");
#ifndef K1F_OVS
#define K1F_OVS 8           // samples per bucket (K1F_NB * K1F_OVS u64 keys are sorted in LDS by one workgroup: 128 KB)
#endif
#define K1F_S (K1F_NB * K1F_OVS)
#ifndef K1F_C
#define K1F_C 1024          // rotations a bucket-sort workgroup holds in LDS
#endif
#ifndef K1F_PT
#define K1F_PT 16384        // rotations per partition tile (8192 -> 16384, round 3: a tile's share of a bucket, 8 indices, is what one write
                            // request of k1f_scatter carries; k1f_scatter / k1f_scan 0.40 / 0.07 -> 0.30 / 0.04 ms per 10^8 bytes)
#endif

// HIP-event timing of K1's main kernels, launch by launch, on the stream they are launched on (bench.py's roofline leg picks
// the kernel with the largest total for the workload at hand).  Filled by the host drivers when enabled.
#define K1_PROF_MAX 8192
#define K1_PROF_CLASSES 8
enum { K1P_BSORT = 0, K1P_RROUND = 1, K1P_DBUILD = 2, K1P_DROUND = 3, K1P_DMED = 4, K1P_DLARGE = 5, K1P_DUPDATE = 6, K1P_TASK = 7 };
struct K1Prof {
    int enabled;
    u32 used;                             // event pairs recorded since the profile was enabled
    u64 elements[K1_PROF_CLASSES];        // elements the launches of a class processed (where the host knows them)
    u32 dbl_runs;                         // k1_dbl_run calls (the entries of their lists are read back from the workspace)
    unsigned char* cls;                   // [K1_PROF_MAX] class of every pair
    hipEvent_t* ev;                       // [2 * K1_PROF_MAX]
};
int k1_prof_enable(K1Prof& p, int on);
int k1_prof_read(K1Prof& p, u32 cls, float* total_ms, u32* launches, u64* elements);
void k1_prof_destroy(K1Prof& p);
// slots are reserved atomically: sub-batches on different streams are driven by different host threads
static inline u32 k1_prof_begin(K1Prof* pr, u32 cls, hipStream_t stream) {
    if (!pr || !pr->enabled) return K1_PROF_MAX;
    const u32 slot = __atomic_fetch_add(&pr->used, 1u, __ATOMIC_RELAXED);
    if (slot >= K1_PROF_MAX) return K1_PROF_MAX;
    pr->cls[slot] = (unsigned char)cls;
    (void)hipEventRecord(pr->ev[2 * slot], stream);
    return slot;
}
static inline void k1_prof_end(K1Prof* pr, u32 slot, hipStream_t stream, u64 elements) {
    if (slot >= K1_PROF_MAX) return;
    (void)hipEventRecord(pr->ev[2 * slot + 1], stream);
    __atomic_fetch_add(&pr->elements[pr->cls[slot]], elements, __ATOMIC_RELAXED);
}

struct K1Buf {
    K1Prof* prof;     // optional
    u32 linear;       // 0: cyclic rotations (bzip2); 1: suffixes with implicit smallest sentinel
    int* SAout;       // linear mode: optional copy of the suffix array [nb][stride]
    const u8* T;      // [nb][tstride]  T_ext[i] = T[i mod n]
    u32* SA;          // [nb][stride]   suffix array (result)
    u32* SB;          // [nb][stride]   ping-pong of the front end; doubling rounds: R, the new rank of every position of a big group
    u32* ISA;         // [nb][stride]   rank (= position of the group head) of every rotation (doubling rounds)
    u32* KA;          // [nb][stride]   scratch: bucket ids of the task levels; keys of large groups (ping)
    u32* KB;          // [nb][stride]   scratch: keys of large groups (pong), then the lengths of their sub-groups at the heads
    u32* HN;          // [nb][hstride]  head bitmap: bit p set = suffix-array position p starts a group (bits at and beyond n are set)
    const u32* nlen;  // [nb]           block lengths
    u32* nfront;      // [nb]           the length the general sort sees: nlen, or 0 where k1_period.hip wrote the suffix array
    u32* per;         // [nb]           smallest period of the block if <= 64 (k1_period.hip), else 0
    u32* ptab;        // [nb][256]      tables of a periodic block
    u32* red;         // [nb]           period p > 64 of a block that is sorted through its first 3 p + (n mod p) bytes (k1_period.hip), else 0
    u32* tileHist;    // [nb][ptiles][K1F_NB]  front end: per-tile bucket counts
    u64* fsplit;      // [nb][K1F_NB]       front end: bucket d holds the keys in [fsplit[d-1], fsplit[d])
    u8* fsub;         // [nb][K1F_NB]       round 6, heavy 8-byte keys: 0 = an ordinary bucket; n = 1 .. 254: the bucket of ONE key, the first of n that share the key's
                      //                    rotations by their NEXT 8 bytes; 255: another one of those n
    u64* fsplit2;     // [nb][K1F_NB]       ... bucket d of such a run holds the rotations whose next 8 bytes are below fsplit2[d] (and not below fsplit2[d - 1])
    u8* fp16;         // [nb][K1F_NB]       ... 1: bucket d holds ONE 16-byte key (sub-splitters x, x + 1: a heavy continuation of the heavy key): beyond LDS it is a group as it stands
    u32* fstart;      // [nb][K1F_NB+1]     front end: first suffix-array position of every bucket
    u32* stats;       // [K1_STATS]
    u32* deepCnt;     // [2 passes][2 classes][8 XCD regions][K1_DEEP_SUB]  entries in each list sub-region of the lane kernels
                      //   (pass 2: what the last refinement round handed over; pass 1 is not filled any more)
    u64* listT[2];    // doubling rounds: descriptors of groups of K1D_GS+1 .. 1024 rotations (cur/next)
    u64* listS[2];    // lane kernels' lists (pairs / groups of 3..8); [0] also the chunks of the doubling rounds' large groups
    u64* listM[2];    // descriptors of groups of 1025 .. K1_MED_MAX rotations
    u64* listL[2];    // ... of more than K1_MED_MAX
    u32 listTCap, listSCap, listMCap, listLCap;
    u64* rlist[2];    // [nb][stride]   entry lists: refinement rounds (in/out), doubling rounds ([0] the round's list, [1] its re-ordered copy)
    u32* rcnt;        // [K1R_MAXR + 1][nb8][K1_RCS]  refinement rounds: entries per round and block (K1_RCNT)
    u32 rstride;      // words per row of dcnt / dchg (and of dtot, dred): 8 x (nb8 / 8 rounded up to 32), indexed by K1_BI
    u32 rnb8;         // blocks rounded up to 8: row length of rcnt (in counters)
    u32* dcnt;        // [K1D_MAXR + 2][rstride]  doubling rounds: list entries per round and block
    u32* dchg;        // [K1D_MAXR + 2][rstride]  != 0: a group of the block split in that round (none: only identical rotations are left)
    u32* dtot;        // [rstride]                positions in unsorted groups per block before the doubling rounds (k1_count_unsorted)
    u32* dbn;         // [K1D_MAXR + 2][4]        per round: descriptors of medium groups (1025..), of large groups, chunks, medium groups (..1024)
    u32* dred;        // [rstride]                != 0: the block was reduced by k1_period.hip (behind dbn: read back together with dtot)
    uint4* btask;     // [K1F_LEVELS][8][btaskCap]  task levels of the front end (k1f_task): (block, position, length, depth | flag), one list per level and XCD (block mod 8)
    u32* bcnt;        // [K1F_LEVELS][8]         tasks per level and XCD
    u32 btaskCap;
    u32 btaskLists;   // task lists in use per level: min(8, blocks of the batch)
    u8* U;            // [nb][stride]   BWT output
    u32* pidx;        // [nb]           origPtr
    u32* hpin;        // host side only: pinned memory for K1's small read-backs (null: pageable), hpinWords u32 long
    u32 hpinWords;
};

// bytes of workspace K1 needs for a batch geometry (everything except T, nlen, U, pidx)
size_t k1_workspace_bytes(const BatchGeom& g);
// carve the workspace; `ws` must be 256-byte aligned
void k1_carve(K1Buf& B, const BatchGeom& g, void* ws);
// enqueue the whole K1 pipeline on `stream`; max_n = largest block length in the batch
int k1_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream);
// k1_front.hip: rotations of every block sorted by their first 8 bytes into B.SA, group heads into B.HN
// iters: 0 = the bucket sort compares 8 bytes, else K1F_KEYB = 16 (cyclic mode with lists); lists: fill the round lists (0: neither - the
// K1-deep tile kernel does that work); purerot_max: the predictor threshold (see k1f_bsort)
int k1_front_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream, u32 iters, u32 lists, u32 purerot_max, u32 carry);
int k1_rounds_run(K1Buf B, const BatchGeom& g, hipStream_t stream, u32 depth0, u32 max_depth, u32 carry);
// k1_dbl.hip: ranks of every rotation from (SA, HN), then list-driven prefix doubling from depth h0 until every group is resolved
//   check_h != 0: from the round with that h on the host looks after every round whether anything is left and stops launching if not
int k1_dbl_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream, u32 h0, u32 check_h);
// k1_period.hip: blocks with a linear period <= 64 get their suffix array from a closed form and leave the general sort
// (nfront[b] = 0); also sets nfront for every other block: runs before the front end.  enable = 0: no block qualifies
int k1_period_run(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream, u32 enable);
// after the sort: the suffix arrays of the blocks with red[b] != 0 from those of their reduced blocks (into SB: k1_finish reads them there)
int k1_period_expand(K1Buf B, const BatchGeom& g, u32 max_n, hipStream_t stream);
#define K1R_STEP 24u           // text bytes a refinement round (k1r_round) takes off every listed rotation
#define K1F_KEYB 16u           // text bytes the bucket sort (k1f_bsort / k1f_task) compares in its one pass (round 5; rounds 3-4: 8, then 12 more per in-bucket iteration)
size_t k1_front_tilehist_words(const BatchGeom& g);   // u32 per block the front end needs in tileHist
