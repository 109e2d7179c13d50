// psa_fill_aux.hip.h -- the small kernels around the fill: FASTA-to-device encoding (symbol ids of A, the
// per-symbol substitution table of a stripe's B) and the boundary chunks of a flagged fill.  Compiled in ONE
// translation unit (psa_fill_score.hip).
#pragma once
#include "psa_args.hip.h"

namespace psa {

// ---- alphabet + table construction (FASTA-to-device encoding) ----
__global__ void alphabet_presence(const uint8_t* a, int64_t n, unsigned* present /*[256]*/) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) present[a[i]] = 1u;
}
// meta[0] = number of distinct symbols; map[byte] = id (in byte order), inv[id] = byte
__global__ void alphabet_compact(const unsigned* present, unsigned* meta, uint8_t* map, uint8_t* inv) {
    if (threadIdx.x != 0) return;
    unsigned k = 0;
    for (int b = 0; b < 256; ++b) {
        if (present[b]) {
            map[b] = (uint8_t)(k < (unsigned)kMaxSym ? k : 0);
            if (k < (unsigned)kMaxSym) inv[k] = (uint8_t)b;
            ++k;
        }
    }
    meta[0] = k;
}
__global__ void encode_rows(const uint8_t* a, int64_t n, const uint8_t* map, uint8_t* ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ids[i] = map[a[i]];
}
__global__ void build_eqtab(const uint8_t* b, int W, const unsigned* meta, const uint8_t* inv, uint32_t* tab,
                            int pitch) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;  // word index inside the row
    const unsigned sym = blockIdx.y;
    if (w >= pitch || sym >= meta[0]) return;   // (rows of residues that do not occur in A are never read: a_ids < meta[0])
    uint32_t word = 0x06060606u;
    if (sym < (unsigned)kMaxSym) {
        const int c0 = (w - kTabPad) * 4;
        const unsigned ch = inv[sym];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + k;
            if (c >= 0 && c < W && b[c] == ch) word += 2u << (8 * k);
        }
    }
    tab[(size_t)sym * pitch + w] = word;
}

// ---- the same for every slot of a multi-pair launch at once (psa_many: BindBatch), three launches instead of a dozen
// stream operations per slot.  blockIdx.y = job.
// bind_stage: zero the regions the fill must find zeroed, copy the residues into the slot's buffers, mark the residues
// of A that occur (present[] is all zero on entry: bind_alphabet leaves it so).
__global__ void bind_stage(const BindBatch B) {
    const BindJob& q = B.job[blockIdx.y];
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int r = 0; r < kBindRegions; ++r) {
        uint4* p = (uint4*)q.clear[r];
        const int64_t cnt = q.clear_bytes[r] / 16;
        for (int64_t i = tid; i < cnt; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (!q.tables) return;
    const uint8_t* a = q.src_a ? q.src_a : q.a;
    const bool copy_a = q.src_a && q.src_a != q.a;
    for (int64_t i = tid; i < q.n; i += stride) {
        const uint8_t ch = a[i];
        if (copy_a) q.a[i] = ch;
        q.alpha[ch] = 1u;
    }
    if (q.src_b && q.src_b != q.b)
        for (int64_t i = tid; i < q.W; i += stride) q.b[i] = q.src_b[i];
}
// bind_alphabet: one workgroup of 256 threads per job, thread = byte value.  meta[0] = number of distinct residues,
// map[byte] = id (in byte order), inv[id] = byte; present[] is zeroed for the next binding.
__global__ void bind_alphabet(const BindBatch B) {
    const BindJob& q = B.job[blockIdx.x];
    if (!q.tables) return;
    __shared__ unsigned wave_count[4];
    unsigned* present = q.alpha;
    unsigned* meta = q.alpha + 256;
    uint8_t* map = (uint8_t*)(meta + 4);
    uint8_t* inv = map + 256;
    const int b = threadIdx.x;
    const bool have = present[b] != 0u;
    const unsigned long long mask = __ballot(have);
    const int lane = b & 63, w = b >> 6;
    if (lane == 0) wave_count[w] = (unsigned)__popcll(mask);
    __syncthreads();
    unsigned before = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
    for (int x = 0; x < w; ++x) before += wave_count[x];
    if (have) { map[b] = (uint8_t)before; inv[before] = (uint8_t)b; }
    if (b == 0) meta[0] = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
    present[b] = 0u;
}
// bind_tables: symbol ids of A's rows and the table rows of the residues that occur (build_eqtab's words).
__global__ void bind_tables(const BindBatch B) {
    const BindJob& q = B.job[blockIdx.y];
    if (!q.tables) return;
    const unsigned* meta = q.alpha + 256;
    const uint8_t* map = (const uint8_t*)(meta + 4);
    const uint8_t* inv = map + 256;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < q.n; i += stride) q.ids[i] = map[q.a[i]];
    const int64_t words = (int64_t)meta[0] * q.pitch;
    for (int64_t e = tid; e < words; e += stride) {
        const int sym = (int)(e / q.pitch), w = (int)(e - (int64_t)sym * q.pitch);
        const int c0 = (w - kTabPad) * 4;
        const unsigned ch = inv[sym];
        uint32_t word = 0x06060606u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + k;
            if (c >= 0 && c < q.W && q.b[c] == ch) word += 2u << (8 * k);
        }
        q.eqtab[e] = word;
    }
}
// one block of kResultWords threads per job: the walk's state, the final triple and the abort word of a slot side by side
__global__ void collect_results(const CollectBatch C) {
    const CollectJob& q = C.job[blockIdx.x];
    const int t = threadIdx.x;
    int32_t v = 0;
    if (t < 24) v = q.state ? ((const int32_t*)q.state)[t] : 0;
    else if (t < 27) v = q.triple[t - 24];
    else if (t == 27) v = (int32_t)*q.abort_flag;
    C.out[(size_t)q.slot * kResultWords + t] = v;
}

// ---- boundary chunks of a flagged fill (psa_stripe_fill_flagged: one launch over all rows of a stripe whose left
// boundary arrives, and whose right boundary leaves, in row chunks while the launch runs) ----
// packed[k][y] (k < 3, y < rows) -> plane[k * n + r0 + y].  Stored past this GPU's caches (system scope, as the fill's
// own boundary accesses: kEdgeScope in psa_fill_v2.hip.h): the fill reads them with loads of the same scope once it has seen the flags, which edge_flags raises in a launch of its own behind this one.
__global__ void edge_scatter(const int32_t* packed, int64_t r0, int64_t rows, int64_t n, int32_t* plane) {
    const int64_t total = 3 * rows, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t k = e / rows, y = e - k * rows;
        __hip_atomic_store(plane + k * n + r0 + y, packed[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void edge_flags(uint32_t* flags, unsigned count) {
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) __hip_atomic_store(flags + e, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// Waits until the fill has raised the flag of EVERY strip of the chunk, then plane[k * n + r0 + y] -> packed[k][y].
// (Strips do not raise their flags in order: a strip with a right boundary raises its flag only after its second,
// off-chain pass over its last superblocks -- run_capture_pass, ~7 us -- while the strip below finishes its main pass
// 3 us after it; a wave slowed in that replay can be overtaken.  So each thread polls flags of its own and the block
// goes on when all of them are up.)  Gives up (leaving packed as it is) when the fill has given up or after 60 s, and
// says so in the stripe's abort word.
__global__ void edge_gather(const int32_t* plane, const uint32_t* done, int64_t r0, int64_t rows, int64_t n, int32_t* packed,
                            unsigned* abort_flag) {
    const int64_t f0 = r0 / 64, f1 = (r0 + rows - 1) / 64;
    const unsigned long long t0 = wall_clock64();
    int good = 1;
    for (int64_t f = f0 + threadIdx.x; f <= f1 && good; f += blockDim.x) {
        while (__hip_atomic_load(done + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
            __builtin_amdgcn_s_sleep(127);
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) || wall_clock64() - t0 > 2 * kSpinTicks) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                good = 0;
                break;
            }
        }
    }
    if (!__syncthreads_and(good)) return;
    const int64_t total = 3 * rows, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int64_t k = e / rows, y = e - k * rows;
        packed[e] = __hip_atomic_load(plane + k * n + r0 + y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace psa
