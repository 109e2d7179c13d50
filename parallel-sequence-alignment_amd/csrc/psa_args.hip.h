// psa_args.hip.h -- kernel argument blocks and launch-shape constants shared by the kernel translation units
// (psa_fill_*.hip, psa_trace.hip, psa_batch.hip) and the host translation unit (psa_hip.hip).  No kernels here:
// the library is built from several translation units (Makefile), each kernel is compiled in exactly one.
#pragma once
#include "psa_device.hip.h"

namespace psa {

struct FillArgs {
    const uint8_t* a;        // all of A (n bytes)
    const uint8_t* b;        // this stripe's columns of B (W bytes)
    int64_t n;               // rows of the whole matrix
    int64_t r0, r1;          // this chunk fills rows r0+1..r1
    int64_t c0;              // global column of the left boundary; stripe = columns c0+1..c0+W
    int W;                   // stripe width (>= 1)
    int Q;                   // blocks of 64 steps per strip = ceil((W+63)/64)
    int Wpad;                // ring pitch (granules per parity)
    int tab_pitch;           // fill v2: words per row of the substitution table (FillArgs2::eqtab); here to keep the block at 200 bytes
    const int32_t* left;     // [3][R] plain T1,T2,T3 at column c0 for rows r0+1..r1; null = column 0
    const int32_t* corner;   // [4] plain triple at (r0, c0) and the tag r0, when left != null and r0 > 0: published by
    int32_t* corner_next;    // the chunk above (its first strip stores {left triple of row r1, tag r1} here), which may
                             // still be running -- chunks of one stripe are launched on alternating streams
    int32_t* right;          // [3][R] plain triple at column c0+W, or null
    const unsigned* left_done;  // stripes of one pair filled by ONE launch: left_done[strip] != 0 once the stripe on the
    unsigned* right_done;       // left has stored that strip's rows of `left`; this stripe raises right_done[strip]
                                // behind its stores to `right` (both null: boundaries cross a kernel boundary)
    int32_t* final_triple;   // [3] plain triple at (n, c0+W), written by the strip holding row n
    u64* ring;               // [2][Wpad] granules: bottom row of strip s lives in parity s&1
    uint8_t* codes;          // winner codes (one byte per cell, psa_device.hip.h) or null
    int2* leftcol;           // c0 > 0 and codes: receives (D, H') (biased) of column c0 for every row, index = global row
    unsigned* ticket;        // strip ticket counter (zeroed per launch)
    unsigned* abort_flag;    // set when a bounded spin expires
    u64* prof;               // optional [strips][4]: start, end (100 MHz ticks), placement; null = off
    u64* ckpt;               // banded mode: [bands][Wpad] granules; the last strip of band b-1 publishes its
    int64_t band_strips;     // bottom row to ckpt[b] instead of the ring and band b's first strip reads it
                             // there (0 = off).  A multiple of the strips per workgroup.
};

// ---- fill v2 (psa_fill_v2.hip.h): launch shape ----
// Dynamic LDS a fill workgroup is launched with and never uses (see fill_strips_v2).  One pair alone: 96 KB, so that
// neither a second fill workgroup nor a traceback walk (70 KB) fits on the CU beside it.  Several pairs in one
// launch: the stripes of ONE pair 70 KB -- still no second fill workgroup (2 x 82 KB > 160 KB), but walks find
// room beside it (82 + 70 KB); independent pairs (fill_strips_v2_dense) 16 KB: two workgroups per CU, limited by
// their registers, and a walk's workgroup beside them where registers allow.
constexpr int kKeepoutAloneBytes = 96 * 1024;
constexpr int kKeepoutManyBytes = 70 * 1024;
constexpr int kKeepoutDenseBytes = 16 * 1024;
constexpr int kFrontWord = 32;  // control block: [0] ticket, [1] abort flag, [32] front (own cache line)
// Bounded waiting inside kernels: wall-clock time (s_memrealtime, 100 MHz), see Spin in psa_fill_v2.hip.h
constexpr unsigned long long kSpinTicks = 30ull * 100000000ull;   // 30 s
constexpr int kNW = 4;         // strips (wavefronts) per workgroup
// The packed kernel (psa_fill_pk.hip.h) has its own constant (Makefile: PKNW), measured on the bench's launches of 20
// pairs with three wavefronts per SIMD: 4 strips per workgroup 3 200 GCUPS, 3: 2 966, 2: 2 835, 6: 2 356, 12: 2 940.
// Four wavefronts land one on each SIMD of a CU; other sizes leave the SIMDs unevenly loaded, and that outweighs the
// hand-overs through HBM granules that larger workgroups save (a launch alone: 30.0 / 31.3 / 31.8 / 37.6 / 35.0 ms).
#ifndef PSA_PK_NW
#define PSA_PK_NW 4
#endif
constexpr int kPkNW = PSA_PK_NW;
constexpr int kMaxSym = 256;   // rows of the substitution table: one per distinct residue of A (only the rows of residues that occur are built)
constexpr int kTabPad = 16;    // table words (8 columns each) of padding in front of a row

struct FillArgs2 {
    FillArgs base;
    const uint8_t* a_ids;     // symbol id (< kMaxSym) of every row of A
    const uint32_t* eqtab;    // [kMaxSym][base.tab_pitch] words, byte c%4 of word kTabPad + c/4 = score(sym, B[c]): a byte, so that
                              // the step adds it with an SDWA byte select and needs no extract
};

// One launch can fill several stripes at once -- of different pairs, or the column stripes of one pair.
// The workgroups draw tickets from one counter and order[ticket] names (job, group of 4 strips).  The host
// lays the order out so that every job advances at the pace of its own dependency chain (independent
// pairs: round-robin; stripes of one pair: stripe g+1 trails stripe g by the time a strip takes) and
// together they keep every CU busy, which one 2^16 pair alone cannot (its chain is n + m steps long
// whatever the hardware).  A workgroup only ever waits for workgroups with lower tickets, which run
// already or have finished: no assumption on dispatch order or residency.
constexpr int kMaxJobs = 20;          // 20 x 200 bytes + 24: the most a 4 KB kernel argument block holds
constexpr int kOrderJobShift = 24;   // order[t] = job << 24 | group
static_assert(sizeof(FillArgs2) == 200 && kMaxJobs * sizeof(FillArgs2) + 24 <= 4096, "FillBatch must stay within the 4 KB argument block");
struct FillBatch {
    FillArgs2 job[kMaxJobs];
    int njobs;
    const uint32_t* order;        // null: one job, group = ticket
    unsigned* ticket;             // launch-wide ticket counter, zeroed before the launch
};

// ---- binding the pairs of a multi-pair launch (psa_fill_aux.hip.h: bind_stage / bind_alphabet / bind_tables): what
// the engine would otherwise queue per slot (two copies, six memsets, four small kernels) as three launches over all slots
constexpr int kBindRegions = 4;
struct BindJob {
    const uint8_t* src_a;    // residues to copy into the slot's own buffers; null: already there (or the slot is only cleared)
    const uint8_t* src_b;
    uint8_t* a;              // the slot's A (n bytes) and B (W bytes)
    uint8_t* b;
    int64_t n;
    int W;
    int pitch;               // words per row of eqtab
    unsigned* alpha;         // present[256] | meta[4] | map[256 bytes] | inv[256 bytes]; present is all zero between launches
    uint8_t* ids;            // symbol id of every row of A
    uint32_t* eqtab;
    void* clear[kBindRegions];             // regions the fill must find zeroed (ring, corner slots, control blocks),
    unsigned clear_bytes[kBindRegions];    // multiples of 16 bytes, 16-byte aligned; 0 = none
    int tables;              // 1: the slot holds a new pair (copy, alphabet, ids, table); 0: only the regions are cleared
    int pad_;
};
struct BindBatch {
    BindJob job[kMaxJobs];
    int njobs;
};
// results of a multi-pair launch gathered into one block per slot (one copy to the host instead of three per slot):
// words 0..23 the walk's state (12 x int64), 24..26 the final triple, 27 the fill's abort word
constexpr int kResultWords = 32;
struct CollectJob { const int64_t* state; const int32_t* triple; const unsigned* abort_flag; int slot; int pad_; };
struct CollectBatch {
    CollectJob job[kMaxJobs];
    int njobs;
    int32_t* out;            // [slots][kResultWords]
};

// ---- traceback walk (replaces the per-cell alignment vectors of simple.cpp:164-211) ----
struct TraceArgs {
    const uint8_t* codes;    // tile of strip 0, block 0 (banded mode: a virtual base, see row_min)
    const uint8_t* a;        // all of A
    const uint8_t* b;        // this stripe's columns of B: b[c] = residue of column c0 + 1 + c
    const int2* leftcol;     // c0 > 0: (D, H') (biased) of column c0, index = global row; null when c0 == 0
    const u64* toprow;       // row_min > 0: granules (D << 2 | V' - D) of row row_min, index = column - c0 - 1
    int64_t c0;              // stripe = columns c0+1..c0+W
    int W, Q;
    int64_t cap;             // capacity of out_ai / out_bj
    int32_t* out_ai;         // columns in walk (reverse) order
    int32_t* out_bj;
    int64_t* state;          // in: i, j, k ; out: i, j, k, done (1 = reached the matrix boundary), len, error
    int64_t row_min;         // banded mode: the codes at hand cover rows > row_min only (0: whole matrix);
                             // `codes` then points row_min/64 strips before the band's first tile
    int64_t n;               // rows of the whole matrix
    const int32_t* start_triple;   // not null: the walk starts at (n, c0 + W) in the table the final pick of this
                                   // triple names (simple.cpp:246-256) instead of at state[0..2]
};
constexpr int kMaxWalks = 20;
struct TraceBatch {          // one wavefront (workgroup) per walk
    TraceArgs job[kMaxWalks];
    int njobs;
};

// ---- batch kernel (psa_batch.hip.h) ----
constexpr int kBatchWaves = 4;      // waves per workgroup
constexpr int kBatchMaxLen = 1024;  // longest B the LDS row buffer holds; longer pairs use psa_score
constexpr int32_t kBatchSkipped = INT32_MIN;   // score slot of a pair the kernel did not align (B longer than max_b)

struct BatchArgs {
    const uint8_t* seqs;
    const int64_t* offsets;
    const int32_t* pairs;
    int64_t npairs;
    int32_t* scores;
    unsigned* next_pair;   // work counter, zeroed before the launch
    int max_b;             // longest second sequence in the batch, rounded up to a multiple of 64
    int no_packed;         // 1: every pair through the int32 path (cross-check of the packed 16-bit path: PSA_BATCH_PACKED=0)
};

// dynamic LDS per wave: row buffer (slot s = (D of column s-65, V of column s-64), as in fill v2),
// B residues with 128 entries of padding in front, and a dump for the lanes that publish nothing
__host__ __device__ inline int batch_row_slots(int max_b) { return max_b + 160; }
// (the B region holds one int per column for a couple of pairs in the packed 16-bit path, bytes in the int32 path)
__host__ __device__ inline int batch_wave_bytes(int max_b) { return batch_row_slots(max_b) * 8 + 4 * (max_b + 256) + 112 * 4; }

}  // namespace psa
