// kmer_table.h -- the open-addressing k-mer table of K-kmer (k <= 31) as device functions: shared by
// kmer_count.hip (every occurrence goes to the table) and kmer_bulk.hip (only what its LDS path hands back does).
// See kmer_count.hip for the reference it restates and for why a key's first slot is a function of its minimizer.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace covest {
namespace kmer {

constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int kMaxProbe = 1 << 16;

// ASCII base -> 2-bit code for a/c/g/t in either case: bits 2:1 give a=0 c=1 t=2 g=3; x ^ (x>>1) swaps g,t.
__device__ __forceinline__ unsigned base_code(unsigned char ch)
{
    const unsigned x = (ch >> 1) & 3u;
    return x ^ (x >> 1);
}

__device__ __forceinline__ unsigned long long slot_of(unsigned long long key, int log2_slots)
{
    return (key * 0x9E3779B97F4A7C15ull) >> (64 - log2_slots); // Fibonacci hashing
}

// A key's FIRST slot: [ line | slot in the line ], a line being 8 slots = 128 bytes.
//   line  a hash of the key's MINIMIZER -- the canonical m-mer of smallest hash among its k - m + 1 = 8 windows
//         (m = k - 7).  Neighbouring windows of a read mostly share their minimizer, so the 64 k-mers a wave inserts
//         at a time fall into a dozen lines instead of 64 unrelated ones;
//   slot  the POSITION of the minimizer inside the key (plus a rotation taken from its hash): the k-mers of one run
//         -- same minimizer, positions 7, 6, 5, ... as the window moves on -- get slots of their own by construction.
// Why: what bounds this kernel is the rate at which the memory side retires scattered 8-byte operations, about 2e10
// load+add pairs a second whatever the table's size (tools/microbench_atomics.hip), and requests of one wave
// instruction that fall into one 128-byte line are retired together (3.8x the rate with 8 lanes per line).
// A key that finds its first slot taken by ANOTHER key does not probe on from there -- the k-mers that differ from
// a genomic one by a substitution mostly share its minimizer and its position, and would walk through the run's
// occupied slots -- but goes to the plain hash of the key and probes linearly from there (table_add).  Slots never
// change their key, so every occurrence of a key takes the same decisions: exact counts.  `rc`: the reverse
// complement of `key` as a 2k-bit code (the function is symmetric in the two up to the mirrored position).
constexpr int kLineLog2 = 3;
constexpr int kLineSlots = 1 << kLineLog2;

__device__ __forceinline__ unsigned long long revcomp_code(unsigned long long x, int k)
{
    unsigned long long rc = 0;
    for (int i = 0; i < k; ++i) {
        rc = (rc << 2) | (3ull - (x & 3ull));
        x >>= 2;
    }
    return rc;
}

__device__ __forceinline__ bool has_first_slot(const KmerTable &t)
{
    return t.k - (kLineSlots - 1) >= 6 && t.log2_slots > kLineLog2 + 4;
}

__device__ __forceinline__ unsigned long long first_slot(unsigned long long key, unsigned long long rc, const KmerTable &t)
{
    const int m = t.k - (kLineSlots - 1);
    const unsigned long long mm = (1ull << (2 * m)) - 1ull;
    unsigned long long best = ~0ull;
    int at = 0;
#pragma unroll
    for (int i = 0; i < kLineSlots; ++i) {
        const unsigned long long a = (key >> (2 * i)) & mm;                    // m-mer i of the key ...
        const unsigned long long b = (rc >> (2 * (kLineSlots - 1 - i))) & mm;  // ... and its reverse complement
        const unsigned long long c = a < b ? a : b;
        const unsigned long long x = (c + 1ull) * 0x9E3779B97F4A7C15ull;
        if (x < best) {
            best = x;
            at = i;
        }
    }
    const unsigned long long h2 = best * 0xD6E8FEB86659FD93ull; // (the minimum of 8 hashes is small: spread it again)
    const unsigned long long line = h2 >> (64 - (t.log2_slots - kLineLog2));
    return (line << kLineLog2) | (unsigned long long)((at + (int)(h2 & 7ull)) & (kLineSlots - 1));
}

// true: the key was counted in slot h (found there, or put there)
__device__ __forceinline__ bool try_slot(const KmerTable &t, unsigned long long h, unsigned long long key,
                                         unsigned long long add)
{
    KmerSlot *slot = t.slots + h;
    unsigned long long cur = __hip_atomic_load(&slot->key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == kEmptyKey)
        cur = atomicCAS(&slot->key, kEmptyKey, key); // returns the previous value
    if (cur == kEmptyKey || cur == key) {
        atomicAdd(&slot->count, add);
        return true;
    }
    return false;
}

__device__ __forceinline__ void table_add(const KmerTable t, unsigned long long key, unsigned long long rc,
                                          unsigned long long add, int *overflow)
{
    if (has_first_slot(t) && try_slot(t, first_slot(key, rc, t), key, add))
        return; // (a second try in the same line before leaving it was measured: no change)
    unsigned long long h = slot_of(key, t.log2_slots);
    for (int probe = 0; probe < kMaxProbe; ++probe) {
        if (try_slot(t, h, key, add))
            return;
        h = (h + 1) & t.mask;
    }
    *overflow = 1;
}


// The window starting at seq[s] of a read of `len` bases as 2-bit codes, little-endian (base i of the window at bits
// 2i) -- that IS the reverse complement's code once complemented -- and mirrored (first base in the highest bits:
// hash_kmer, bin/kmer_hist.py:18-23).  Four bases per (unaligned) 32-bit load where the read has them.
__device__ __forceinline__ void window_codes(const unsigned char *__restrict__ seq, int64_t s, int64_t len, int k,
                                             unsigned long long &h, unsigned long long &rc)
{
    const int n_words = (k + 3) >> 2; // 32-bit words of 4 bases that cover a window
    const unsigned long long kmask = k < 32 ? (1ull << (2 * k)) - 1ull : ~0ull;
    h = 0;
    rc = 0;
    if (s + 4 * n_words <= len) {
        unsigned long long le = 0;
        for (int j = 0; j < n_words; ++j) {
            unsigned w;
            __builtin_memcpy(&w, seq + s + 4 * j, 4);
            unsigned x = (w >> 1) & 0x03030303u;
            x ^= (x >> 1) & 0x01010101u;
            le |= (unsigned long long)((x * 0x01041040u) >> 24) << (8 * j);
        }
        le &= kmask;
        rc = ~le & kmask;
        unsigned long long r = __brevll(le);
        r = ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
        h = r >> (64 - 2 * k);
    } else { // the last windows of a read: byte by byte (a word would reach past the read's end)
        for (int i = 0; i < k; ++i) {
            const unsigned long long c = base_code(seq[s + i]);
            h = (h << 2) | c;                  // hash_kmer, :18-23 (rehash :26-31 yields the same window code)
            rc |= (3ull - c) << (2 * i);       // reverse complement, built back to front
        }
    }
}

// from the little-endian code of a window (base i at bits 2i, masked to 2k bits): the window's own code and its
// reverse complement's
__device__ __forceinline__ void codes_from_le(unsigned long long le, int k, unsigned long long &h, unsigned long long &rc)
{
    const unsigned long long kmask = k < 32 ? (1ull << (2 * k)) - 1ull : ~0ull;
    rc = ~le & kmask;
    unsigned long long r = __brevll(le);
    r = ((r & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((r & 0x5555555555555555ull) << 1);
    h = r >> (64 - 2 * k);
}

} // namespace kmer
} // namespace covest
