// The resident calling pipeline (pf_call.hip has the overview), device side: K-SITES: k_call_sites.
#include "pf_call_kernels.hpp"

namespace pf_call {

// ---------------------------------------------------------------------------------------------------------------------
// K-SITES

// UnitigColors::contains(um, colour) (bifrost/src/ColorSet.cpp:776-823) for the mapping [dist, dist + n_km) of unitig u: the colour
// on every one of those k-mers
__device__ inline bool colour_contains(const SiteArgs &a, uint32_t u, uint32_t c, uint32_t dist, uint32_t n_km) {
    if (colour_in(a.full, a.cwords, u, c)) return true;
    for (uint32_t e = a.part_first[u]; e < a.part_first[u + 1]; ++e) {
        if (a.part_colour[e] != c) continue;
        const uint64_t *bits = a.part_bits + a.part_word[e];
        for (uint32_t i = dist; i < dist + n_km; ++i)
            if (!((bits[i >> 6] >> (i & 63)) & 1)) return false;
        return true;
    }
    return false;
}

// cdbg.findUnitig(s, 0, len) of src/CCDBG.cpp:3251, 3390 followed by UnitigColors::contains on that mapping, for one site string:
// its first k-mer lies on one of the bubble's unitigs (a k-mer occurs once in the graph, in one orientation); the mapping is extended
// along that unitig while the characters agree (CompactedDBG.tcc:3815-3837, CompressedSequence.cpp:497-520).  Leaves the set of
// colours present on every k-mer of the mapping in out[0 .. a.cwords) (bit c % 64 of word c / 64); false when no unitig of the
// bubble holds the first k-mer.
__device__ inline bool colours_of_string(const SiteArgs &a, const char *sp, uint32_t lp, const uint32_t *walk, uint32_t n_walk, uint64_t *out) {
    const int k = a.k;
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    for (uint32_t w = 0; w < a.cwords; ++w) out[w] = 0;
    if (lp < (uint32_t)k) return false;
    auto code = [](char ch) -> int { return ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : ch == 'T' ? 3 : -1; };
    uint64_t head = 0;
    for (int i = 0; i < k; ++i) {
        const int b = code(sp[i]);
        if (b < 0) return false;   // (a gap character never matches a unitig)
        head = (head << 2) | (uint64_t)b;
    }
    const uint64_t rhead = rc_kmer(head, k);
    for (uint32_t q = 0; q < n_walk; ++q) {
        const uint32_t u = walk[q] >> 1;
        const uint32_t Lu = a.len[u];
        const uint64_t *w = a.seq + a.off[u];
        auto base_at = [&](uint32_t x) -> int { return (int)((w[x >> 5] >> (62 - 2 * (x & 31))) & 3u); };
        uint64_t y = 0, word = 0;
        int hit = 0;   // 1 forward, 2 reverse complement
        uint32_t p0 = 0;
        // (std::string::find(head) over the whole unitig first, then find(rhead): a unitig holds at most one of the two)
        for (uint32_t pos = 0; pos < Lu && hit != 1; ++pos) {
            if ((pos & 31) == 0) word = w[pos >> 5];
            y = ((y << 2) | ((word >> (62 - 2 * (pos & 31))) & 3u)) & kmask;
            if (pos + 1 < (uint32_t)k) continue;
            if (y == head) { hit = 1; p0 = pos + 1 - (uint32_t)k; }
            else if (y == rhead && !hit) { hit = 2; p0 = pos + 1 - (uint32_t)k; }
        }
        if (!hit) continue;
        uint32_t dist, n_km;
        if (hit == 1) {
            uint32_t jn = (uint32_t)k;
            while (jn < lp && p0 + jn < Lu && code(sp[jn]) == base_at(p0 + jn)) ++jn;
            n_km = jn - (uint32_t)k + 1;
            dist = p0;
        } else {
            long ps = (long)p0 + k - 1;
            uint32_t jn = 0;
            while (jn < lp && ps >= 0 && code(sp[jn]) == 3 - base_at((uint32_t)ps)) { ++jn; --ps; }
            n_km = jn - (uint32_t)k + 1;
            dist = p0 - (n_km - 1);
        }
        for (uint32_t c = 0; c < a.n_colors; ++c)
            if (colour_contains(a, u, c, dist, n_km)) out[c >> 6] |= 1ull << (c & 63);
        return true;
    }
    return false;
}

template <bool COLORED>
__global__ __launch_bounds__(64, 4) void k_call_sites(SiteArgs a) {
    const int lane = lane_id();
    const uint32_t KS = a.ks;
    const uint32_t C = COLORED ? a.n_colors : 1;
    const int k = a.k;
    uint8_t *scr = a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave;
    // per row (rows_cap rows: 256, more when a bubble of the batch has more walks): appended characters, final string, and the per-row scalars
    const size_t RC = a.rows_cap;
    char *app = reinterpret_cast<char *>(scr);
    char *fin = app + RC * KS;
    uint32_t *flen = reinterpret_cast<uint32_t *>(fin + RC * KS);
    uint32_t *at = flen + RC;
    uint32_t *rank = at + RC;
    uint8_t *dup = reinterpret_cast<uint8_t *>(rank + RC);
    uint8_t *sok = dup + RC;
    double *mean = reinterpret_cast<double *>(sok + RC);
    // colored, per row: the colours its string's mapping carries in full and the colours whose range test it passed (sets of CW
    // 64-bit words), its mean per colour, and whether findUnitig found nothing for it
    const uint32_t CW = COLORED ? a.cwords : 1;
    uint64_t *cmask = reinterpret_cast<uint64_t *>(mean + RC);   // [rows_cap][CW]
    uint64_t *cokm = cmask + RC * CW;                            // [rows_cap][CW]
    double *cmean = reinterpret_cast<double *>(cokm + RC * CW);  // [rows_cap][C]
    uint8_t *cnf = reinterpret_cast<uint8_t *>(cmean + RC * C);  // [rows_cap]
    const uint64_t kmask = (1ull << (2 * k)) - 1;
    const uint32_t n_branching = a.cnt->n_branching;
    unsigned long long my_strings = 0;   // (added to the batch's count once, at the end)
    unsigned long long pk[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long pk0 = a.prof ? wall_clock64() : 0;
    auto sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    };
    // No shared queue head and one bump of the value pool per SV_CHUNK values instead of one per bubble: two atomics per bubble on
    // two addresses -- 60 000 per launch, served one after the other by the L2 -- were what the launch took, whatever its
    // wavefronts did in between.  Wavefront w takes bubbles w, w + grid, ...
    constexpr uint32_t SV_CHUNK = 1024;
    unsigned long long chunk_at = 0;
    uint32_t chunk_left = 0;
    // (... for all but the last two rounds, which are handed out one bubble at a time: the wavefronts finish together)
    const uint32_t rounds = n_branching / gridDim.x;
    const uint32_t n_static = rounds > 2 ? (rounds - 2) * gridDim.x : 0;
    uint32_t q_static = blockIdx.x;
    for (;;) {
        uint32_t q;
        if (q_static < n_static) {
            q = q_static;
            q_static += gridDim.x;
        } else {
            q = 0;
            if (lane == 0) q = atomicAdd(&a.cnt->sites_next, 1u);
            q = n_static + read_lane(q, 0);
            if (q >= n_branching) break;
        }
        const unsigned long long pa = a.prof ? wall_clock64() : 0;
        const uint32_t j = a.blist[q];
        const pf_bubble_result r = a.res[j];
        if (r.n_rows == 0 || r.n_rows == 0xFFFFFFFFu) continue;
        ++pk[5];
        const uint32_t R = r.n_rows, L = r.n_cols;
        const char *rows = a.otext + r.rows_off;
        // values: one slot per allele group and site, plus the site's sum
        uint32_t n_val = 0;
        for (uint32_t si = 0; si < r.n_sites; ++si) n_val += C * (uint32_t)a.osites[r.site_off + si].maxnum + 1;
        unsigned long long v0 = 0;
        if (n_val > chunk_left) {   // (wave-uniform)
            const uint32_t take = n_val > SV_CHUNK ? n_val : SV_CHUNK;
            if (lane == 0) v0 = atomicAdd(&a.cnt->sv_head, (unsigned long long)take);
            chunk_at = ((unsigned long long)read_lane((uint32_t)(v0 >> 32), 0) << 32) | read_lane((uint32_t)v0, 0);
            chunk_left = take;
        }
        v0 = chunk_at;
        chunk_at += n_val;
        chunk_left -= n_val;
        if (lane == 0) a.sv_off[j] = v0;
        const bool room = v0 + n_val <= a.sv_cap;
        uint32_t err = 0, n_strings = 0;
        uint32_t indel = 0;
        unsigned long long vcur = v0;
        if (a.prof) pk[1] += wall_clock64() - pa;
        for (uint32_t si = 0; si < r.n_sites && !err; ++si) {
            const unsigned long long pb = a.prof ? wall_clock64() : 0;
            const pf_bubble_site sr = a.osites[r.site_off + si];
            const uint8_t *grp = a.ogroups + r.group_off + (uint64_t)si * R;
            const uint32_t site = sr.col;
            const uint32_t maxnum = sr.maxnum;
            // A site that is no indel in a bubble that has met none so far -- most sites -- takes k raw columns of every row
            // (src/CDBG.cpp:1559-1596): equally long strings over {-, A, C, G, T}, k <= 31.  Lane p holds row p's string as
            // order-preserving 3-bit codes in two registers (comparing them = comparing the strings) next to the 2-bit k-mer
            // the probe wants; ranks, duplicates and the groups' sums go through lane reads instead of the scratch arrays,
            // whose every access is a step in a chain of dependent loads.  Same decisions, same order of the additions.
            const long plain_from = (long)site - k + 1;
            if (!COLORED && !sr.is_indel && indel == 0 && plain_from >= 0 && (uint64_t)plain_from + (uint64_t)k <= L && k <= 31 && R <= WAVE) {
                const uint32_t p = (uint32_t)lane;
                const bool mine = p < R;
                uint64_t hi = 0, lo = 0, km = 0;
                uint32_t g = 0;
                if (mine) {
                    // (all 32 bytes asked for at once -- a loop of loads would wait for each in turn; the row pool is allocated
                    // with slack, so the bytes past the k-th exist)
                    unsigned char cs[32];
                    __builtin_memcpy(cs, rows + (size_t)p * L + plain_from, 32);
#pragma unroll
                    for (int x = 0; x < 31; ++x) {   // (constant indices: cs stays in registers)
                        if (x >= k) continue;
                        const char ch = (char)cs[x];
                        const uint64_t c3 = ch == '-' ? 0 : ch == 'A' ? 1 : ch == 'C' ? 2 : ch == 'G' ? 3 : 4;
                        hi = (hi << 3) | (lo >> 61);
                        lo = (lo << 3) | c3;
                        km = (km << 2) | (ch == 'A' ? 0 : ch == 'C' ? 1 : ch == 'G' ? 2 : 3);
                    }
                    km &= kmask;
                    g = grp[p];
                }
                bool d = false;
                uint32_t rk = 0;
                for (uint32_t o = 0; o < R; ++o) {
                    const uint64_t ohi = ((uint64_t)read_lane((uint32_t)(hi >> 32), (int)o) << 32) | read_lane((uint32_t)hi, (int)o);
                    const uint64_t olo = ((uint64_t)read_lane((uint32_t)(lo >> 32), (int)o) << 32) | read_lane((uint32_t)lo, (int)o);
                    const uint32_t og = read_lane(g, (int)o);
                    if (!mine || o == p || og != g) continue;
                    if (ohi == hi && olo == lo) { if (o < p) d = true; }
                    else if (ohi < hi || (ohi == hi && olo < lo)) ++rk;
                }
                const unsigned long long pc = a.prof ? wall_clock64() : 0;
                pk[2] += pc - pb;
                bool miss = false, okp = true;
                double mn = 0.0;
                if (mine && !d) {
                    uint64_t sum = 0;
                    if (!a.tab_exact) {
                        uint32_t cnt;
                        if (!canonical_count(a.tab, a.mask, km, k, cnt, a.one_strand != 0)) miss = true;
                        else if (cnt > a.low && cnt < a.up) sum = cnt;
                        else okp = false;   // src/CDBG.cpp:45-50
                    }
                    mn = (double)sum;   // (one k-mer: the mean is its count)
                }
                n_strings += (uint32_t)__popcll(__ballot(mine && !d));
                const unsigned long long pd = a.prof ? wall_clock64() : 0;
                pk[3] += pd - pc;
                // group coverages in set order; the first string out of range drops the site (src/CDBG.cpp:1527-1551) -- and ends the
                // loop: a string BEHIND it is never looked up, so a k-mer of such a string that is in no database ends nothing
                // (the reference's exit sits inside readCov, :52-56).  Only a missing k-mer the walk reaches is the run's end.
                bool ok = true, fatal = false;
                double total = 0.0;
                for (uint32_t gi = 0; gi < maxnum; ++gi) {
                    double tc = 0.0;
                    if (ok) {
                        uint32_t last = 0;
                        bool have_last = false;
                        for (;;) {
                            const bool cand = mine && g == gi + 1 && !d && (!have_last || rk > last);
                            const unsigned long long m = __ballot(cand);
                            if (!m) break;
                            const uint32_t best = read_lane(wave_min_u32(cand ? rk : 0xFFFFFFFFu), 0);
                            const int bp = __ffsll((long long)__ballot(cand && rk == best)) - 1;   // (distinct strings of a group: distinct ranks)
                            const uint32_t verdict = read_lane(miss ? 2u : (okp ? 1u : 0u), bp);
                            if (verdict == 2) { fatal = true; ok = false; break; }
                            if (!verdict) { ok = false; break; }
                            const uint64_t mb = (uint64_t)__double_as_longlong(mn);
                            tc += __longlong_as_double((long long)(((uint64_t)read_lane((uint32_t)(mb >> 32), bp) << 32) | read_lane((uint32_t)mb, bp)));
                            last = best;
                            have_last = true;
                        }
                        if (ok) total += tc;
                    }
                    if (lane == 0 && room) a.sv[vcur + gi] = tc;
                }
                if (fatal) { err = 2; break; }
                if (lane == 0 && room) a.sv[vcur + maxnum] = total;
                if (lane == 0) a.osites[r.site_off + si].pad_ = ok ? 1 : 0;
                vcur += maxnum + 1;
                if (a.prof) pk[4] += wall_clock64() - pd;
                continue;
            }
            uint32_t napp = 0;
            if (sr.is_indel) {
                // every path: the next non-gap character at / after the site, again and again until the characters just appended
                // are not all equal (src/CDBG.cpp:1474-1493)
                for (uint32_t p = lane; p < R; p += WAVE) at[p] = site;
                sync();
                // A row that ends in gaps runs out here: the reference's `substr(pos, 1)` at pos == size() is the EMPTY string --
                // nothing is appended to that row, '\0' joins the set of characters (so the loop ends unless every row ran out at
                // once) and pos moves on to size() + 1, where the next substr throws and the reference terminates.  A row that ran
                // out keeps at[p] = L + 1 and has one character fewer than the others.
                for (;;) {
                    uint32_t bad = 0;
                    for (uint32_t base = 0; base < R; base += WAVE) {
                        const uint32_t p = base + lane;
                        bool e = false;
                        if (p < R) {
                            uint32_t x = at[p];
                            if (x > L) e = true;   // (every row ran out in the round before: std::out_of_range in the reference)
                            else {
                                while (x < L && rows[(size_t)p * L + x] == '-') ++x;
                                if (napp < KS) app[(size_t)p * KS + napp] = x < L ? rows[(size_t)p * L + x] : '\0';
                                at[p] = x + 1;
                            }
                        }
                        if (__ballot(e)) bad = 1;
                    }
                    if (napp >= KS) {   // (still growing: ask for the most a string can reach -- a row's characters and its raw columns)
                        bad = 16;
                        if (lane == 0) atomicMax(&a.cnt->ks_need, 2 * L + (uint32_t)k + 2);
                    }
                    sync();
                    if (bad) { err = bad == 16 ? 16 : 4; break; }
                    const char first = app[napp];
                    bool differ = false;
                    for (uint32_t base = 0; base < R; base += WAVE) {
                        const uint32_t p = base + lane;
                        if (__ballot(p < R && app[(size_t)p * KS + napp] != first)) differ = true;
                    }
                    ++napp;
                    if (differ) break;
                }
                if (err) break;
            }
            // the k-length string of every path around the site (src/CDBG.cpp:1494-1525, 1559-1596)
            uint32_t row_err = 0;
            for (uint32_t base = 0; base < R; base += WAVE) {
                const uint32_t p = base + lane;
                uint32_t e = 0;
                if (p < R) {
                    const char *row = rows + (size_t)p * L;
                    char *out = fin + (size_t)p * KS;
                    uint32_t n = 0;
                    auto push = [&](char c) { if (n < KS) out[n] = c; ++n; };
                    if (sr.is_indel) {
                        const uint32_t napp_p = at[p] > L ? napp - 1 : napp;   // (a row that ran out in the last round)
                        const long need = (long)k - (long)napp_p;
                        if (indel == 0) {
                            // substr(site - k + n, k - n): a start below zero or past the row throws; a negative count is npos
                            const long from = (long)site - k + (long)napp_p;
                            if (from < 0 || (uint64_t)from > L) e = 4;
                            else {
                                const uint32_t take = need < 0 ? (uint32_t)(L - (uint64_t)from) : (uint32_t)std::min<uint64_t>((uint64_t)need, L - (uint64_t)from);
                                for (uint32_t x = 0; x < take; ++x) push(row[from + x]);
                                for (uint32_t x = 0; x < napp_p; ++x) push(app[(size_t)p * KS + x]);
                            }
                        } else {
                            uint32_t c = 0;
                            for (uint32_t x = 0; x < site && x < L; ++x) c += row[x] != '-';
                            if (need < 0 || (long)c < need) {
                                for (uint32_t x = 0; x < site && x < L; ++x)
                                    if (row[x] != '-') push(row[x]);
                                for (uint32_t x = 0; x < napp_p; ++x) push(app[(size_t)p * KS + x]);
                                for (uint32_t x = at[p]; n < (uint32_t)k; ++x) {
                                    if (x >= L) { e = 4; break; }
                                    if (row[x] != '-') push(row[x]);
                                }
                            } else {
                                // the last `need` non-gap characters before the site
                                uint32_t skip = c - (uint32_t)need;
                                for (uint32_t x = 0; x < site && x < L; ++x) {
                                    if (row[x] == '-') continue;
                                    if (skip) { --skip; continue; }
                                    push(row[x]);
                                }
                                for (uint32_t x = 0; x < napp_p; ++x) push(app[(size_t)p * KS + x]);
                            }
                        }
                    } else if (indel > 0) {
                        uint32_t c = 0;
                        for (uint32_t x = 0; x <= site && x < L; ++x) c += row[x] != '-';
                        if (c < (uint32_t)k) {
                            for (uint32_t x = 0; x <= site && x < L; ++x)
                                if (row[x] != '-') push(row[x]);
                            for (uint32_t x = site + 1; n < (uint32_t)k; ++x) {
                                if (x >= L) { e = 4; break; }
                                if (row[x] != '-') push(row[x]);
                            }
                        } else {
                            uint32_t skip = c - (uint32_t)k;
                            for (uint32_t x = 0; x <= site && x < L; ++x) {
                                if (row[x] == '-') continue;
                                if (skip) { --skip; continue; }
                                push(row[x]);
                            }
                        }
                    } else {
                        const long from = (long)site - k + 1;
                        if (from < 0 || (uint64_t)from > L) e = 4;
                        else {
                            const uint32_t take = (uint32_t)std::min<uint64_t>((uint64_t)k, L - (uint64_t)from);
                            for (uint32_t x = 0; x < take; ++x) push(row[from + x]);
                        }
                    }
                    if (n > KS) { e = 16; atomicMax(&a.cnt->ks_need, n); }
                    flen[p] = n;
                }
                if (__ballot(e == 4)) row_err |= 4;
                if (__ballot(e == 16)) row_err |= 16;
            }
            if (sr.is_indel) ++indel;
            sync();
            const unsigned long long pc = a.prof ? wall_clock64() : 0;
            pk[2] += pc - pb;
            if (row_err) { err = (row_err & 4) ? 4 : 16; break; }
            // distinct strings per allele group in std::set order, their coverage (readCov(string), src/CDBG.cpp:29-60)
            for (uint32_t base = 0; base < R; base += WAVE) {
                const uint32_t p = base + lane;
                bool miss = false;
                if (p < R) {
                    const char *sp = fin + (size_t)p * KS;
                    const uint32_t lp = flen[p];
                    const uint8_t g = grp[p];
                    bool d = false;
                    uint32_t rk = 0;
                    for (uint32_t o = 0; o < R; ++o) {
                        if (o == p || grp[o] != g) continue;
                        const char *so = fin + (size_t)o * KS;
                        const uint32_t lo = flen[o], lm = lo < lp ? lo : lp;
                        uint32_t x = 0;
                        while (x < lm && so[x] == sp[x]) ++x;
                        int cmp;  // so <=> sp
                        if (x < lm) cmp = (unsigned char)so[x] < (unsigned char)sp[x] ? -1 : 1;
                        else cmp = lo < lp ? -1 : (lo > lp ? 1 : 0);
                        if (cmp == 0) { if (o < p) d = true; }
                        else if (cmp < 0) ++rk;
                    }
                    dup[p] = d;
                    rank[p] = rk;   // counts duplicates of smaller strings as well: only the order of the ranks matters
                    uint8_t ok = 1;
                    double mn = 0.0;
                    if (COLORED) {
                        if (!d) {
                            // readCov(string, low, up, c) (src/CCDBG.cpp:89-122) for every colour: one look at the joined table per k-mer
                            uint64_t *okw = cokm + (size_t)p * CW;
                            for (uint32_t w = 0; w < CW; ++w) okw[w] = C - 64 * w >= 64 ? ~0ull : ((1ull << (C - 64 * w)) - 1);
                            uint64_t *cs = reinterpret_cast<uint64_t *>(cmean + (size_t)p * C);
                            for (uint32_t c = 0; c < C; ++c) cs[c] = 0;
                            StringWindow win;
                            for (uint32_t c0 = 0; c0 < lp; ++c0) {
                                const uint64_t x = win.push(sp[c0], kmask, (uint32_t)k);
                                if (c0 + 1 < (uint32_t)k) continue;
                                const uint32_t *sa, *sb;
                                colored_slots(a.ctab, x, k, a.c_one_strand != 0, sa, sb);
                                for (uint32_t w = 0; w < CW; ++w) {
                                    uint64_t okm = okw[w];
                                    const uint32_t c_end = C - 64 * w >= 64 ? 64 * w + 64 : C;
                                    for (uint32_t c = 64 * w; c < c_end; ++c) {
                                        if (!((okm >> (c & 63)) & 1) || a.unread[c]) continue;   // (a colour never looked up: (0, true))
                                        const uint32_t cnt = ctab_count(sa, sb, c);
                                        if (cnt != CTAB_MISSING && cnt > a.clow[c] && cnt < a.cup[c]) cs[c] += cnt;
                                        else { cs[c] = 0; okm &= ~(1ull << (c & 63)); }   // missing or outside (low, up): (0, false), :105-117
                                    }
                                    okw[w] = okm;
                                }
                            }
                            for (uint32_t c = 0; c < C; ++c) cmean[(size_t)p * C + c] = (double)cs[c] / (double)((uint64_t)lp - (uint64_t)k + 1);
                            const uint64_t wf = a.walk_off[j];
                            // (findUnitig finds nothing: fatal in the reference IF the walk below reaches this string)
                            cnf[p] = colours_of_string(a, sp, lp, a.walk_pool + (wf & ((1ull << 40) - 1)), (uint32_t)(wf >> 40), cmask + (size_t)p * CW) ? 0 : 1;
                        }
                    } else if (!d) {
                        uint64_t sum = 0;
                        if (!a.tab_exact) {
                            StringWindow win;
                            for (uint32_t c = 0; c < lp; ++c) {
                                const uint64_t x = win.push(sp[c], kmask, (uint32_t)k);
                                if (c + 1 >= (uint32_t)k) {
                                    uint32_t cnt;
                                    if (!canonical_count(a.tab, a.mask, x, k, cnt, a.one_strand != 0)) { miss = true; break; }
                                    if (cnt > a.low && cnt < a.up) sum += cnt;
                                    else { sum = 0; ok = 0; break; }  // src/CDBG.cpp:45-50
                                }
                            }
                        }
                        mn = (double)sum / (double)((uint64_t)lp - (uint64_t)k + 1);
                    }
                    sok[p] = miss ? 2 : ok;   // (2: a k-mer in no database -- the reference's exit, if the walk below reaches this string)
                    mean[p] = mn;
                }
                n_strings += (uint32_t)__popcll(__ballot(p < R && !dup[p]));
            }
            sync();
            const unsigned long long pd = a.prof ? wall_clock64() : 0;
            pk[3] += pd - pc;
            bool fatal = false;
            if (COLORED) {
                // src/CCDBG.cpp:3236-3339, 3374-3475: per allele group its strings in set order; a colour the string's mapping carries
                // in full adds the string's mean to [colour][group]; a failed range test of such a colour, or a colour no string
                // carries, drops the site.  Lane l keeps the sums of colour 64 w + l, one word w of the colour sets after the other (the
                // walk over the strings does not depend on the word, so every round takes the same steps and ends the same way).
                bool ok = true, all_seen = true;
                for (uint32_t w = 0; w < CW && !fatal; ++w) {
                    const uint32_t c = 64 * w + (uint32_t)lane;
                    uint64_t seen = 0;
                    ok = true;
                    for (uint32_t gi = 0; gi < maxnum; ++gi) {
                        double tc = 0.0;
                        uint32_t last = 0;
                        bool have_last = false;
                        while (ok) {
                            uint32_t best = 0xFFFFFFFFu, bp = 0;
                            for (uint32_t p = 0; p < R; ++p) {
                                if (grp[p] != gi + 1 || dup[p]) continue;
                                const uint32_t rk = rank[p];
                                if (have_last && rk <= last) continue;
                                if (rk < best) { best = rk; bp = p; }
                            }
                            if (best == 0xFFFFFFFFu) break;
                            if (cnf[bp]) { fatal = true; ok = false; break; }
                            const uint64_t *mw = cmask + (size_t)bp * CW, *kw = cokm + (size_t)bp * CW;
                            const uint64_t m = mw[w];
                            seen |= m;
                            for (uint32_t x = 0; x < CW && ok; ++x) ok = !(mw[x] & ~kw[x]);
                            if (!ok) break;
                            if (c < C && ((m >> lane) & 1)) tc += cmean[(size_t)bp * C + c];
                            last = best;
                            have_last = true;
                        }
                        if (c < C && room) a.sv[vcur + (uint64_t)c * maxnum + gi] = tc;
                    }
                    all_seen = all_seen && seen == (C - 64 * w >= 64 ? ~0ull : ((1ull << (C - 64 * w)) - 1));
                }
                if (fatal) { err = 64; break; }
                const bool valid = ok && all_seen;
                if (lane == 0 && room) a.sv[vcur + (uint64_t)C * maxnum] = valid ? 1.0 : 0.0;
                if (lane == 0) a.osites[r.site_off + si].pad_ = valid ? 1 : 0;
                vcur += (uint64_t)C * maxnum + 1;
                sync();
                if (a.prof) pk[4] += wall_clock64() - pd;
                continue;
            }
            // group coverages in set order; the first string out of range drops the site (src/CDBG.cpp:1527-1551)
            bool ok = true;
            double total = 0.0;
            for (uint32_t gi = 0; gi < maxnum; ++gi) {
                double tc = 0.0;
                if (ok) {
                    // rows of this group, ascending rank: wave-uniform selection of the next smallest rank
                    uint32_t last = 0;
                    bool have_last = false;
                    for (;;) {
                        uint32_t best = 0xFFFFFFFFu, bp = 0;
                        for (uint32_t p = 0; p < R; ++p) {
                            if (grp[p] != gi + 1 || dup[p]) continue;
                            const uint32_t rk = rank[p];
                            if (have_last && rk <= last) continue;
                            if (rk < best) { best = rk; bp = p; }
                        }
                        if (best == 0xFFFFFFFFu) break;
                        if (sok[bp] == 2) { fatal = true; ok = false; break; }
                        if (!sok[bp]) { ok = false; break; }
                        tc += mean[bp];
                        last = best;
                        have_last = true;
                    }
                    if (ok) total += tc;
                }
                if (lane == 0 && room) a.sv[vcur + gi] = tc;
            }
            if (fatal) { err = 2; break; }
            if (lane == 0 && room) a.sv[vcur + maxnum] = total;
            if (lane == 0) a.osites[r.site_off + si].pad_ = ok ? 1 : 0;
            vcur += maxnum + 1;
            sync();
            if (a.prof) pk[4] += wall_clock64() - pd;
        }
        if (lane == 0 && err) atomicOr(&a.cnt->err, err);
        my_strings += n_strings;
    }
    if (lane == 0 && my_strings) atomicAdd(&a.cnt->site_strings, my_strings);
    if (a.prof && lane == 0) {
        pk[0] = wall_clock64() - pk0;
        for (int x = 0; x < 6; ++x) a.prof[(size_t)blockIdx.x * 6 + x] = pk[x];
    }
}

// the forms pf_call.hip launches
template __global__ void k_call_sites<true>(SiteArgs);
template __global__ void k_call_sites<false>(SiteArgs);

}  // namespace pf_call
