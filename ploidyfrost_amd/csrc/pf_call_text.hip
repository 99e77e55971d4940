// The resident calling pipeline (pf_call.hip has the overview), device side: K-TEXT and the other text the device writes: k_call_has, k_call_format, k_call_totals, k_sb_count, k_sb_format, k_format_doubles.
#include "pf_call_kernels.hpp"

namespace pf_call {

// ---------------------------------------------------------------------------------------------------------------------
// K-TEXT
// indel_len[indel - 1] as the callers print it (src/CDBG.cpp:1310, 1565; src/CCDBG.cpp:3034, 3315, 3450).  An indel run still open
// at the last column never has its length pushed (compareStrPair closes a run only on a later column, src/SeqAlign.cpp:56-157), so
// for that last indel site the reference reads one element past the vector: heap garbage that changes from run to run, or a null
// dereference when the vector is empty -- rows ending in gaps, i.e. gap-friendly scores only.  Undefined there; defined here (and in
// the oracle, pfo::indel_len_at) as the length of the open run: columns - the site's column.
__device__ __forceinline__ uint32_t open_run_len(const uint32_t *ilen, uint32_t i, uint32_t n_ilen, uint32_t n_cols, uint32_t col) {
    return i < n_ilen ? ilen[i] : n_cols - col;
}


// computeCramerVCoefficient (src/CCDBG.cpp:330-366) on rows ca and cb of a [colour][allele] coverage matrix given as val(colour, allele)
template <class Val>
__device__ inline double cramer_v_dev(const Val &val, uint32_t ca, uint32_t cb, uint32_t n_alleles) {
#pragma clang fp contract(off)
    double n = 0, nA = 0, nB = 0, chi = 0;
    uint32_t count = 0;
    for (uint32_t i = 0; i < n_alleles; ++i) {
        const double A = val(ca, i), B = val(cb, i), p = A + B;
        nA += A;
        nB += B;
        n = n + p;
        if (p != 0) ++count;
    }
    if ((count & 255u) < 2) return 0;   // (the reference counts in a uint8_t)
    for (uint32_t i = 0; i < n_alleles; ++i) {
        const double A = val(ca, i), B = val(cb, i), p = A + B;
        if (p == 0) continue;
        const double exA = nA * p / n, exB = nB * p / n;
        const double dA = A - exA, dB = B - exB;
        chi += dA * dA / exA;   // pow(x, 2) is x * x, correctly rounded, in glibc as here
        chi += dB * dB / exB;
    }
    return sqrt(chi / n);
}
// the largest over all colour pairs (:2964-2970, 3285-3291); std::max keeps its first argument when the second is NaN
template <class Val>
__device__ inline double max_cramer_v_dev(const Val &val, uint32_t n_colors, uint32_t n_alleles) {
    double c = 0;
    for (uint32_t ci = 0; ci + 1 < n_colors; ++ci)
        for (uint32_t cj = ci + 1; cj < n_colors; ++cj) {
            const double v = cramer_v_dev(val, ci, cj, n_alleles);
            c = c < v ? v : c;
        }
    return c;
}

template <bool W>
struct Row {  // one output stream position: a pointer when writing, a byte count when measuring
    char *p;
    uint32_t n;
    __device__ inline void put(char c) { if (W) *p++ = c; else ++n; }
};

// two streams that receive the same characters (a frequency row goes to its arity's file and to allele_frequency.txt): formatted
// once, stored twice -- copying the first stream's bytes back out of memory made every character wait for the store before it
template <bool W>
struct Tee {
    Row<W> &a, &b;
    __device__ inline void put(char c) { a.put(c); b.put(c); }
};

__global__ void k_call_has(const pf_bubble_result *__restrict__ res, uint32_t nb, uint32_t *__restrict__ has) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < nb) has[j] = (res[j].n_rows != 0 && res[j].n_rows != 0xFFFFFFFFu) ? 1u : 0u;
}

// The write pass stages the four large streams in LDS: consecutive lanes hold consecutive bubbles, so a wavefront's text in a
// stream is one contiguous span of the output; it is formatted into LDS and copied out by consecutive lanes (whole 64-byte
// segments per store).  A span that does not fit its stage (long rows) is written directly, byte by byte, as before.
// Stages of 13 KB a wavefront and 168 registers (launch bounds: three wavefronts a SIMD; 171 and two before): six blocks a CU
// instead of four -- a launch 0.194 -> 0.177 ms, a step 20.4 -> 20.0 ms at configs[2]; stages of 9.7 KB with 128 registers (four a
// SIMD) measure no better (0.181 ms).
constexpr uint32_t FMT_STAGE[4] = {8192, 1536, 1536, 2048};   // alignseq, allele_frequency, bifre, bicov
constexpr int FMT_STAGED_STREAM[4] = {1, 0, 2, 6};

template <bool W, bool COLORED>
__global__ __launch_bounds__(FMT_BLOCK, W ? 3 : 4) void k_call_format(FmtArgs a) {
    const uint32_t jj = blockIdx.x * blockDim.x + threadIdx.x;   // index inside the text batch (sizes / offsets)
    const uint32_t j = a.j0 + jj;                                // ... inside the aligned batch (results, numbering, site values)
    unsigned long long allele[4] = {0, 0, 0, 0}, core_cov = 0, core_num = 0;
    __shared__ __attribute__((aligned(16))) char s_stage[W ? (FMT_BLOCK / 64) * (FMT_STAGE[0] + FMT_STAGE[1] + FMT_STAGE[2] + FMT_STAGE[3]) : 4];
    bool staged[4] = {false, false, false, false};
    uint64_t span0[4] = {0, 0, 0, 0};
    uint32_t span_len[4] = {0, 0, 0, 0};
    char *stage[4] = {nullptr, nullptr, nullptr, nullptr};
    char *cp_dst = nullptr;          // this lane's bubble: where row 0's characters go, where the rows lie, their length and number,
    const char *cp_src = nullptr;    // and the distance from one row's characters to the next row's in the output
    uint32_t cp_L = 0, cp_R = 0, cp_step = 0;
    if (W) {
        const size_t stride = (size_t)a.nb + 1;
        const uint32_t w_first = jj & ~63u;
        if (w_first < a.nb) {   // (wave-uniform)
            const uint32_t w_end = w_first + 64 < a.nb ? w_first + 64 : a.nb;
            char *base = s_stage + (threadIdx.x >> 6) * (FMT_STAGE[0] + FMT_STAGE[1] + FMT_STAGE[2] + FMT_STAGE[3]);
            uint32_t acc = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (alignseq's stage holds the wavefront's packed records instead of its text when the stream leaves packed)
                const int st_ = (q == 0 && a.packed) ? S_PACK : FMT_STAGED_STREAM[q];
                span0[q] = a.offs[st_ * stride + w_first];
                span_len[q] = (uint32_t)(a.offs[st_ * stride + w_end] - span0[q]);
                staged[q] = span_len[q] <= FMT_STAGE[q];
                stage[q] = base + acc;
                acc += FMT_STAGE[q];
            }
        }
    }
    if (jj < a.nb) {
        const size_t stride = (size_t)a.nb + 1;
        Row<W> s_all{nullptr, 0}, s_aln{nullptr, 0}, s_fre[4], s_cov[4];
        for (int x = 0; x < 4; ++x) { s_fre[x] = Row<W>{nullptr, 0}; s_cov[x] = Row<W>{nullptr, 0}; }
        if (W) {
            s_all.p = a.out[0] + (a.offs[0 * stride + jj] - a.offs[0 * stride]);
            if (!a.packed) s_aln.p = a.out[1] + (a.offs[1 * stride + jj] - a.offs[1 * stride]);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                s_fre[x].p = a.out[2 + x] + (a.offs[(2 + x) * stride + jj] - a.offs[(2 + x) * stride]);
                s_cov[x].p = a.out[6 + x] + (a.offs[(6 + x) * stride + jj] - a.offs[(6 + x) * stride]);
            }
            if (staged[0] && !a.packed) s_aln.p = stage[0] + (a.offs[1 * stride + jj] - span0[0]);
            if (staged[1]) s_all.p = stage[1] + (a.offs[0 * stride + jj] - span0[1]);
            if (staged[2]) s_fre[0].p = stage[2] + (a.offs[2 * stride + jj] - span0[2]);
            if (staged[3]) s_cov[0].p = stage[3] + (a.offs[6 * stride + jj] - span0[3]);
        }
        const pf_bubble_result r = a.res[j];
        if (r.n_rows != 0 && r.n_rows != 0xFFFFFFFFu) {
            const CallTask &t = a.ct[a.kept[a.t0 + j]];
            const uint32_t R = r.n_rows, L = r.n_cols;
            const char *rows = a.otext + r.rows_off;
            const uint64_t my_vc = a.vc_base + a.vc[j] - (a.mt ? 1 : 0);   // fetch_add(1) returns the old value (src/CDBG.cpp:2056)
            char *fre_start[4] = {s_fre[0].p, s_fre[1].p, s_fre[2].p, s_fre[3].p};
            const uint32_t fre_n0[4] = {s_fre[0].n, s_fre[1].n, s_fre[2].n, s_fre[3].n};
            if (W && a.packed) {
                // alignseq, packed: this lane writes its bubble's header; the rows are packed by the whole wavefront further down
                char *rec = staged[0] ? stage[0] + (a.offs[(size_t)S_PACK * stride + jj] - span0[0])
                                      : a.out[S_PACK] + alnpack_index_bytes(a.nb) + (a.offs[(size_t)S_PACK * stride + jj] - a.offs[(size_t)S_PACK * stride]);
                const uint64_t vc64 = my_vc;
                const uint32_t h[4] = {t.u + 1, (t.exit_ov >> 1) + 1, L, R | (t.strict ? 0x80000000u : 0u)};
                __builtin_memcpy(rec, &vc64, 8);
                __builtin_memcpy(rec + 8, h, 16);
                cp_dst = rec + ALNPACK_HEADER; cp_src = rows; cp_L = L; cp_R = R;
            }
            // alignseq: var_count, strict flag, entrance id, exit id, aligned row (src/CDBG.cpp:1259, 1428)
            for (uint32_t p = 0; p < R && !(W && a.packed); ++p) {
                char *const row_start = s_aln.p;
                put_uint(s_aln, my_vc);
                s_aln.put('\t'); s_aln.put(t.strict ? '1' : '0'); s_aln.put('\t');
                put_uint(s_aln, (uint64_t)t.u + 1);
                s_aln.put('\t');
                put_uint(s_aln, (uint64_t)(t.exit_ov >> 1) + 1);
                s_aln.put('\t');
                // the aligned row itself is copied by the whole wavefront further down (every row of a bubble has the same prefix, so
                // row p's characters start p * (prefix + L + 1) behind row 0's): a lane copying its own rows byte by byte touches 64
                // different lines per load instruction -- the copies were what the write pass waited for
                if (W) { if (p == 0) { cp_dst = s_aln.p; cp_src = rows; cp_L = L; cp_R = R; } s_aln.p += L; }
                else s_aln.n += L;
                s_aln.put('\n');
                if (W && p == 0) cp_step = (uint32_t)(s_aln.p - row_start);
            }
            core_cov = (unsigned long long)t.core_mean;
            core_num = 1;
            const pf_bubble_site *sites = a.osites + r.site_off;
            const uint32_t *ilen = a.oilen + r.ilen_off;
            const size_t usize = a.len[t.u], esize = a.len[t.exit_ov >> 1];
            const uint32_t ns = r.n_sites;
            uint32_t indel = 0;
            uint64_t vcur = t.strict ? 0 : a.sv_off[j];
            for (uint32_t i = 0; i < ns; ++i) {
                const pf_bubble_site sr = sites[i];
                const uint8_t *grp = a.ogroups + r.group_off + (uint64_t)i * R;
                // distance to the neighbouring sites / unitig ends (src/CDBG.cpp:1279-1298)
                uint32_t vd;
                if (i == 0) {
                    if (ns != 1) vd = (uint32_t)std::min((size_t)(uint32_t)(sites[1].col - sites[0].col - 1), usize);
                    else vd = (uint32_t)std::min(usize, esize);
                } else if (i == ns - 1) {
                    vd = (uint32_t)std::min((size_t)(uint32_t)(sites[i].col - sites[i - 1].col - 1), esize);
                } else {
                    vd = std::min((uint32_t)(sites[i].col - sites[i - 1].col - 1), (uint32_t)(sites[i + 1].col - sites[i].col - 1));
                }
                const uint32_t maxnum = sr.maxnum;
                if (sr.is_indel) ++indel;  // counted even when the site is dropped below (src/CDBG.cpp:1526)
                if (COLORED) {
                    const uint32_t C = a.n_colors;
                    const double *cvals = nullptr;
                    if (!t.strict) {
                        cvals = a.sv + vcur;
                        vcur += (uint64_t)C * maxnum + 1;
                        if (!sr.pad_) continue;   // a string failed a colour's range test, or some colour covers no string (:3292-3300)
                    }
                    // strict: the [colour][path] matrix of the scan, again from K-COV-C's results (an entry = the mean coverage of the
                    // path's unitig in a colour that has it in full, else 0), paths as sorted there
                    uint32_t wu[4] = {0, 0, 0, 0}, lk[4] = {1, 1, 1, 1};
                    if (t.strict)
                        for (uint32_t p = 0; p < R && p < 4; ++p) {
                            wu[p] = t.inner[p] >> 1;
                            lk[p] = a.len[wu[p]] - (uint32_t)a.k + 1;
                        }
                    auto m_at = [&](uint32_t c, uint32_t p) -> double {
                        const uint32_t q = p < 4 ? p : 3;
                        return colour_in(a.full, a.cwords, wu[q], c) ? (double)a.ccov_sum[(size_t)c * a.N + wu[q]] / (double)lk[q] : 0.0;
                    };
                    auto gc_at = [&](uint32_t c, uint32_t x) -> double {   // coverage of allele group x in colour c
                        if (!t.strict) return cvals[(size_t)c * maxnum + x];
                        double tc = 0.0;
                        for (uint32_t p = 0; p < R; ++p)
                            if ((uint32_t)grp[p] - 1 == x) tc += m_at(c, p);
                        return tc;
                    };
                    const double coefficient = t.strict ? max_cramer_v_dev(m_at, C, R) : max_cramer_v_dev(gc_at, C, maxnum);
                    for (uint32_t c = 0; c < C; ++c) {
                        uint32_t n_res = 0;
                        double sum = 0;
                        for (uint32_t x = 0; x < maxnum; ++x) {
                            const double v = gc_at(c, x);
                            if (v > 0.0) { ++n_res; sum += v; }
                        }
                        if (n_res < 2) continue;
                        const int ar = (int)n_res - 2;
                        Row<W> cov = ar == 0 ? s_cov[0] : ar == 1 ? s_cov[1] : ar == 2 ? s_cov[2] : s_cov[3];
                        Row<W> fre = ar == 0 ? s_fre[0] : ar == 1 ? s_fre[1] : ar == 2 ? s_fre[2] : s_fre[3];
                        const bool filed = ar <= 3;
                        for (uint32_t x = 0; x < maxnum; ++x) {
                            const double v = gc_at(c, x);
                            if (!(v > 0.0)) continue;
                            if (filed) { put_double(cov, v); cov.put('\t'); }
                            const double fr = v / sum;
                            if (filed) {
                                Tee<W> both{fre, s_all};
                                put_double(both, fr);
                                both.put('\n');
                            } else {
                                put_double(s_all, fr);
                                s_all.put('\n');
                            }
                        }
                        if (filed) {
                            put_uint(cov, c);
                            cov.put('\t');
                            cov.put(t.strict ? '1' : '0'); cov.put('\t');
                            if (sr.is_indel) put_uint(cov, open_run_len(ilen, indel - 1, r.n_indel_len, r.n_cols, sr.col));
                            else cov.put('0');
                            cov.put('\t');
                            put_uint(cov, my_vc);
                            cov.put('\t');
                            put_uint(cov, ns);
                            cov.put('\t');
                            put_double(cov, coefficient);
                            cov.put('\t');
                            put_uint(cov, vd);
                            cov.put('\t'); cov.put('\n');
                            ++allele[ar];
                            if (ar == 0) { s_cov[0] = cov; s_fre[0] = fre; }
                            else if (ar == 1) { s_cov[1] = cov; s_fre[1] = fre; }
                            else if (ar == 2) { s_cov[2] = cov; s_fre[2] = fre; }
                            else { s_cov[3] = cov; s_fre[3] = fre; }
                        }
                    }
                    continue;
                }
                double denom;
                const double *vals = nullptr;
                if (t.strict) {
                    denom = t.cov_sum;
                } else {
                    vals = a.sv + vcur;
                    vcur += maxnum + 1;
                    if (!sr.pad_) continue;
                    denom = vals[maxnum];
                }
                const int ar = (int)maxnum - 2;  // file of this arity, if 0..3
                Row<W> cov = ar == 0 ? s_cov[0] : ar == 1 ? s_cov[1] : ar == 2 ? s_cov[2] : s_cov[3];
                Row<W> fre = ar == 0 ? s_fre[0] : ar == 1 ? s_fre[1] : ar == 2 ? s_fre[2] : s_fre[3];
                const bool filed = ar >= 0 && ar <= 3;
                for (uint32_t x = 0; x < maxnum; ++x) {
                    double tc;
                    if (t.strict) {
                        tc = 0.0;
                        for (uint32_t p = 0; p < R; ++p)
                            if ((uint32_t)grp[p] - 1 == x) tc += t.cov[p < 4 ? p : 3];
                    } else {
                        tc = vals[x];
                    }
                    if (filed) { put_double(cov, tc); cov.put('\t'); }
                    // the frequency row: the arity's fre file (2..5 alleles) and allele_frequency.txt -- there in site order, or,
                    // in the -t > 1 format, grouped by arity at the end of the bubble (src/CDBG.cpp:2158-2162)
                    const double fr = tc / denom;
                    if (filed) {
                        if (!a.mt) {
                            Tee<W> both{fre, s_all};
                            put_double(both, fr);
                            both.put('\n');
                        } else {
                            put_double(fre, fr);
                            fre.put('\n');
                        }
                    } else if (!a.mt) {
                        put_double(s_all, fr);
                        s_all.put('\n');
                    }
                }
                if (filed) {
                    cov.put(t.strict ? '1' : '0'); cov.put('\t');
                    if (sr.is_indel) put_uint(cov, open_run_len(ilen, indel - 1, r.n_indel_len, r.n_cols, sr.col));
                    else cov.put('0');
                    cov.put('\t');
                    put_uint(cov, my_vc);
                    cov.put('\t');
                    put_uint(cov, ns);
                    cov.put('\t');
                    put_uint(cov, vd);
                    cov.put('\t'); cov.put('\n');
                    ++allele[ar];
                    if (ar == 0) { s_cov[0] = cov; s_fre[0] = fre; }
                    else if (ar == 1) { s_cov[1] = cov; s_fre[1] = fre; }
                    else if (ar == 2) { s_cov[2] = cov; s_fre[2] = fre; }
                    else { s_cov[3] = cov; s_fre[3] = fre; }
                }
            }
            if (a.mt) {
                // allfre << bifre_info << trifre_info << tetrafre_info (<< pentafre_info only in the strict branch, :2162 vs :2550)
                const int n_ar = t.strict ? 4 : 3;
                for (int x = 0; x < n_ar; ++x) {
                    if (W) { for (char *c = fre_start[x]; c < s_fre[x].p; ++c) *s_all.p++ = *c; }
                    else s_all.n += s_fre[x].n - fre_n0[x];
                }
            }
        }
        if (!W) {
            a.sizes[0 * stride + jj] = s_all.n;
            a.sizes[1 * stride + jj] = s_aln.n;
            a.sizes[(size_t)S_PACK * stride + jj] = (a.packed && s_aln.n) ? ALNPACK_HEADER + r.n_rows * alnpack_row_bytes(r.n_cols) : 0u;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                a.sizes[(2 + x) * stride + jj] = s_fre[x].n;
                a.sizes[(6 + x) * stride + jj] = s_cov[x].n;
            }
        }
    }
    if (W) {
        {   // the aligned rows, bubble after bubble, by all lanes
            unsigned long long todo = __ballot(cp_R != 0);
            const int lane = lane_id();
            while (todo) {
                const int b = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const uint32_t Lb = read_lane(cp_L, b), Rb = read_lane(cp_R, b), step = read_lane(cp_step, b);
                const uint64_t d64 = (uint64_t)(uintptr_t)cp_dst, s64 = (uint64_t)(uintptr_t)cp_src;
                char *dst = reinterpret_cast<char *>((uintptr_t)(((uint64_t)read_lane((uint32_t)(d64 >> 32), b) << 32) | read_lane((uint32_t)d64, b)));
                const char *src = reinterpret_cast<const char *>((uintptr_t)(((uint64_t)read_lane((uint32_t)(s64 >> 32), b) << 32) | read_lane((uint32_t)s64, b)));
                if (a.packed) {
                    // eight characters = three bytes (pf_alnpack.hpp).  A lane per character: consecutive lanes read consecutive bytes
                    // of the row, the eight lanes of a group OR their 3-bit codes together (three exchanges), and the first three
                    // lanes of the group store one byte each -- into the LDS stage when the wavefront's records fit it
                    const uint32_t rb = alnpack_row_bytes(Lb);
                    for (uint32_t p = 0; p < Rb; ++p)
                        for (uint32_t x0 = 0; x0 < Lb; x0 += WAVE) {
                            const uint32_t x = x0 + (uint32_t)lane;
                            uint32_t v = x < Lb ? alnpack_code(src[(size_t)p * Lb + x]) << (3 * (lane & 7)) : 0u;
                            v |= (uint32_t)__shfl_xor((int)v, 1, WAVE);
                            v |= (uint32_t)__shfl_xor((int)v, 2, WAVE);
                            v |= (uint32_t)__shfl_xor((int)v, 4, WAVE);
                            const uint32_t g = x >> 3, byte = (uint32_t)lane & 7;
                            if (byte < 3 && (g << 3) < Lb) dst[(size_t)p * rb + 3 * (size_t)g + byte] = (char)(v >> (8 * byte));
                        }
                    continue;
                }
                for (uint32_t p = 0; p < Rb; ++p)
                    for (uint32_t x = (uint32_t)lane; x < Lb; x += WAVE) dst[(size_t)p * step + x] = src[(size_t)p * Lb + x];
            }
        }
        if (a.packed && jj < a.nb && (jj % ALNPACK_GROUP == 0 || jj + 1 == a.nb)) {
            // the index of the piece: where the text and the records of every 256th bubble begin, and where both end
            const size_t stride = (size_t)a.nb + 1;
            char *idx = a.out[S_PACK];
            auto entry = [&](uint64_t g, uint32_t at) {
                const uint64_t e[2] = {a.offs[1 * stride + at] - a.offs[1 * stride], a.offs[(size_t)S_PACK * stride + at] - a.offs[(size_t)S_PACK * stride]};
                __builtin_memcpy(idx + 16 + 16 * g, e, 16);
            };
            if (jj % ALNPACK_GROUP == 0) entry(jj / ALNPACK_GROUP, jj);
            if (jj + 1 == a.nb) {
                const uint64_t n_groups = ((uint64_t)a.nb + ALNPACK_GROUP - 1) / ALNPACK_GROUP;
                const uint64_t head[2] = {n_groups, ALNPACK_GROUP};
                __builtin_memcpy(idx, head, 16);
                entry(n_groups, a.nb);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const size_t stride = (size_t)a.nb + 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!staged[q]) continue;
            const bool pk = q == 0 && a.packed;
            const int st_ = pk ? S_PACK : FMT_STAGED_STREAM[q];
            char *dst = a.out[st_] + (pk ? alnpack_index_bytes(a.nb) : 0) + (span0[q] - a.offs[st_ * stride]);
            // four bytes per lane and step (the stage is word-aligned in LDS; the span lies where it lies in the stream: global
            // memory takes the unaligned word), the last one to three bytes singly
            const uint32_t n_words = span_len[q] >> 2;
            for (uint32_t x = lane_id(); x < n_words; x += WAVE) {
                const uint32_t w = reinterpret_cast<const uint32_t *>(stage[q])[x];
                __builtin_memcpy(dst + 4 * (size_t)x, &w, 4);
            }
            for (uint32_t x = (n_words << 2) + lane_id(); x < span_len[q]; x += WAVE) dst[x] = stage[q][x];
        }
    }
    if (!W) {
        // counters: one atomic per wave and counter
        unsigned long long v[7] = {allele[0], allele[1], allele[2], allele[3], core_cov, core_num, 0};
        for (int x = 0; x < 6; ++x) {
            const unsigned long long s = wave_sum_u64(v[x]);
            // (allele[4], core_cov, core_num lie one after the other: the host zeroes them as six words as well)
            if (lane_id() == 0 && s) atomicAdd(reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(a.cnt) + offsetof(CallCounters, allele)) + x, s);
        }
        if (jj == a.nb) {
            const size_t stride = (size_t)a.nb + 1;
            for (int s = 0; s < N_INT; ++s) a.sizes[s * stride + a.nb] = 0;
        }
    }
}

__global__ void k_call_totals(const uint64_t *__restrict__ offs, const uint32_t *__restrict__ sizes, uint32_t nb, uint64_t *__restrict__ totals) {
    const int s = threadIdx.x;
    const size_t stride = (size_t)nb + 1;
    if (s < N_INT) totals[s] = offs[s * stride + nb] - offs[s * stride];
    (void)sizes;
}

// ---------------------------------------------------------------------------------------------------------------------
// O1, first file: the rows of <outpre>_super_bubble.txt (src/CDBG.cpp:222-252; colored rule src/CCDBG.cpp:2106-2132) from the
// state on the device: one row per open endpoint side in unitig order, numbered from 1.

__device__ inline uint32_t sb_rows_of(const SbArgs &a, uint32_t u, bool &p_row, bool &m_row) {
    const uint8_t f = a.flags[u];
    p_row = m_row = false;
    if ((f & 3) == 0) return 0;
    if (a.colored) { p_row = a.plus[u] != 0; m_row = a.minus[u] != 0; }   // an open unitig lists every side whose partner is set, self included
    else { p_row = (f & B_PLUS) != 0; m_row = (f & B_MINUS) != 0; }
    return (uint32_t)p_row + (uint32_t)m_row;
}

__global__ void k_sb_count(SbArgs a, uint32_t *cnt) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u > a.N) return;
    bool p, m;
    cnt[u] = u < a.N ? sb_rows_of(a, u, p, m) : 0u;
}

constexpr uint32_t SB_STAGE = 4096;   // bytes of LDS per wavefront for its rows (64 unitigs, at most two rows of < 50 bytes each)

template <bool W>
__global__ __launch_bounds__(256) void k_sb_format(SbArgs a) {
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    // write pass: a wavefront's rows are one contiguous span of the file; staged in LDS and copied out by consecutive lanes
    __shared__ __attribute__((aligned(16))) char s_stage[W ? 4 * SB_STAGE : 4];
    char *stage = s_stage + (threadIdx.x >> 6) * SB_STAGE;
    uint64_t span0 = 0;
    uint32_t span_len = 0;
    bool staged = false;
    if (W) {
        const uint32_t w_first = u & ~63u;
        if (w_first < a.N) {
            const uint32_t w_end = w_first + 64 < a.N ? w_first + 64 : a.N;
            span0 = a.offs[w_first];
            span_len = (uint32_t)(a.offs[w_end] - span0);
            staged = span_len <= SB_STAGE;
        }
    }
    if (u < a.N) {
        bool rows[2];
        const uint32_t n = sb_rows_of(a, u, rows[0], rows[1]);
        Row<W> o{W ? (staged ? stage + (a.offs[u] - span0) : a.out + a.offs[u]) : nullptr, 0};
        if (n) {
            const uint8_t f = a.flags[u];
            uint64_t nb = (uint64_t)a.row_base[u] + a.first_id;
            for (int side = 0; side < 2; ++side) {
                if (!rows[side]) continue;
                const bool ps = side == 0;
                put_uint(o, nb++);
                o.put('\t');
                put_uint(o, (uint64_t)u + 1);
                o.put('\t'); o.put(ps ? '+' : '-'); o.put('\t');
                put_uint(o, ps ? a.plus[u] : a.minus[u]);
                o.put('\t'); o.put((f & (ps ? B_STRICT_P : B_STRICT_M)) ? '1' : '0');
                o.put('\t'); o.put((f & (ps ? B_COMPLEX_P : B_COMPLEX_M)) ? '1' : '0');
                o.put('\n');
            }
        }
        if (!W) a.sizes[u] = o.n;
    } else if (u == a.N && !W) {
        a.sizes[u] = 0;
    }
    if (W && staged) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        char *dst = a.out + span0;
        const uint32_t n_words = span_len >> 2;   // (in words, as K-TEXT's stages leave)
        for (uint32_t x = lane_id(); x < n_words; x += WAVE) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(stage)[x];
            __builtin_memcpy(dst + 4 * (size_t)x, &w, 4);
        }
        for (uint32_t x = (n_words << 2) + lane_id(); x < span_len; x += WAVE) dst[x] = stage[x];
    }
}


__global__ void k_format_doubles(const double *__restrict__ x, uint64_t n, char *__restrict__ out, uint8_t *__restrict__ len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BufSink s{out + i * 32};
    put_double(s, x[i]);
    len[i] = (uint8_t)(s.p - (out + i * 32));
}

// K-NIB.  The nine numeric streams are written in sixteen characters -- the digits, '.', tab, newline, '-', 'e', '+' -- and half of
// what a pass sends over PCIe is these streams: a thread takes sixteen characters of a stream's text (as K-TEXT left it in the
// slab) and leaves eight bytes, first character in the low nibble.  A stream with any other character ("nan", "inf": a frequency
// of 0 / 0) sets its bit of the flag word, and the caller fetches that stream as text.  HBM-bound, 1.5 bytes a character.
__global__ __launch_bounds__(256) void k_text_nibbles(NibArgs a) {
    const uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= a.unit0[NIB_STREAMS]) return;
    int s = 0;
#pragma unroll
    for (int x = 1; x < NIB_STREAMS; ++x) s += u >= a.unit0[x] ? 1 : 0;
    const uint64_t at = (u - a.unit0[s]) * 16;
    const char *p = a.src[s] + at;
    const uint64_t left = a.len[s] - at;
    char c[16];
    if (left >= 16) {
        memcpy(c, p, 16);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) c[j] = (uint64_t)j < left ? p[j] : '0';
    }
    uint64_t w = 0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const unsigned ch = (unsigned char)c[j], d = ch - '0';
        unsigned n = d;
        n = ch == '.' ? 10u : n;
        n = ch == '\t' ? 11u : n;
        n = ch == '\n' ? 12u : n;
        n = ch == '-' ? 13u : n;
        n = ch == 'e' ? 14u : n;
        n = ch == '+' ? 15u : n;
        bad = bad || n > 15u;
        w |= (uint64_t)(n & 15u) << (4 * j);
    }
    *reinterpret_cast<uint64_t *>(a.dst[s] + (u - a.unit0[s]) * 8) = w;
    if (bad) atomicOr(a.flag, 1u << a.stream[s]);
}

// the forms pf_call.hip launches
template __global__ void k_call_format<false, false>(FmtArgs);
template __global__ void k_call_format<false, true>(FmtArgs);
template __global__ void k_call_format<true, false>(FmtArgs);
template __global__ void k_call_format<true, true>(FmtArgs);
template __global__ void k_sb_format<false>(SbArgs);
template __global__ void k_sb_format<true>(SbArgs);

}  // namespace pf_call
