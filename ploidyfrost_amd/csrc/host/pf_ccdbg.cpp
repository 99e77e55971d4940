// pfh::ColoredUnitigSet and pfh::CCDBG: the colored (multi-sample) front end, reference src/CCDBG.cpp.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <deque>
#include <mutex>
#include <set>
#include <stdexcept>
#include <thread>

#include "pf_cdbg_impl.hpp"
#include "pf_parallel.hpp"

namespace pfh {

// ---- colored graph + CCDBG (reference src/CCDBG.cpp) -------------------------------------------------
bool ColoredUnitigSet::read(const std::string &graphfile, const std::string &colorfile, size_t nb_threads, bool verbose) {
    if (verbose) printf("ColoredCDBG::read(): Reading graph.\n");
    // abundant k-mers (unitig numbering) are decided by the CCDBG constructor with the device (K-MINZ); on the rare graph
    // whose order changes there, the colour sets are read again for the final order
    if (!graph.load_gfa(graphfile, err, true)) return false;
    if (verbose) printf("ColoredCDBG::read(): Reading colors.\n");
    colorfile_ = colorfile;
    color_threads_ = (unsigned)std::max<size_t>(nb_threads, 1);
    return colors.load(colorfile, graph, color_threads_, err);
}

bool ColoredUnitigSet::reload_colors() { return colors.load(colorfile_, graph, color_threads_, err); }

CCDBG::CCDBG(ColoredUnitigSet &graph, const size_t &complexsize, double &m, double &d, double &g, std::string kmc_db_list,
             const size_t &thread, int device, bool quiet)
    : CDBG(graph.graph, complexsize, m, d, g, device, quiet, NoCounts{}), cg_(graph) {
    if (status_) return;
    if (cg_.graph.n_abundant && !cg_.reload_colors()) { fail(PF_ERR_ARG, "CCDBG::CCDBG():Error: " + cg_.err); return; }
    col_ = &cg_.colors;
    st_.col = col_;
    const uint32_t C = cg_.colors.n_colors;
    if (C > PF_MAX_COLORS_TABLE) { fail(PF_ERR_ARG, "CCDBG::CCDBG():Error: more colours than the device table holds"); return; }
    // the colour gate of the accept commit goes to the device with the graph: the commits run there (pf_replay_device).  Colour sets
    // are (C + 63) / 64 words per unitig, on the host and on the device alike: the reference has no limit on the number of colours
    // (src/CCDBG.cpp:2759-2853 loops over getNbColors()), the joined table's PF_MAX_COLORS_TABLE is the only one here.
    {
        const int st = pf_replay_set_colours(ctx_, C, cg_.colors.full_mask.data(), cg_.colors.size_total.data(), cg_.colors.n_full_enc.data());
        colours_on_device_ = st == PF_OK;
    }
    if (!kmc_db_list.empty()) {
        // src/CCDBG.cpp:13-43: one database name per line, one line per colour
        FILE *f = fopen(kmc_db_list.c_str(), "r");
        if (!f) { fail(PF_ERR_ARG, "CCDBG::CCDBG():Error: Open kmc database name file error"); return; }
        std::vector<std::string> names;
        {
            std::string cur;
            int ch;
            while ((ch = fgetc(f)) != EOF) {
                if (ch == '\n') { names.push_back(cur); cur.clear(); }
                else cur.push_back((char)ch);
            }
            if (!cur.empty()) names.push_back(cur);
            fclose(f);
        }
        names.resize(C);  // missing lines read as empty names, which fail to open below as in the reference
        std::vector<KmcRecords> dbs(C);
        std::vector<std::string> errs(C);
        std::vector<int> bad(C, 0);
        parallel_chunks(C, 1, (unsigned)std::max<size_t>(thread, 1), [&](size_t c, size_t, size_t) { bad[c] = !dbs[c].load(names[c], errs[c]); });
        for (uint32_t c = 0; c < C; ++c) {
            if (bad[c]) { fail(PF_ERR_ARG, "CCDBG::CCDBG():Error: Open kmc database " + names[c] + " error (" + errs[c] + ")"); return; }
            if ((int)dbs[c].k != g_.k) { fail(PF_ERR_ARG, "CCDBG::CCDBG():Error: k of kmc database " + names[c] + " differs from the graph's"); return; }
            if (!quiet_) printf("CCDBG::CCDBG(): kmc database %s initialized\n", names[c].c_str());
        }
        // every database is decoded on the device (K-KMC) from its mapped record file; the joint table is built from those arrays
        std::vector<uint64_t *> pk(C, nullptr);
        std::vector<uint32_t *> pc(C, nullptr);
        std::vector<uint64_t> n(C), mn(C), mx(C);
        std::vector<int> both(C);
        int st = PF_OK;
        for (uint32_t c = 0; c < C && st == PF_OK; ++c) {
            st = pf_kmc_decode(ctx_, dbs[c].records, dbs[c].total, dbs[c].suffix_bytes, dbs[c].counter_size, dbs[c].lut.data(), dbs[c].n_lut(),
                               dbs[c].lut_prefix_len, dbs[c].k, &pk[c], &pc[c]);
            n[c] = dbs[c].total;
            mn[c] = dbs[c].min_count;
            mx[c] = dbs[c].max_count;
            both[c] = dbs[c].both_strands;
        }
        if (st == PF_OK) st = pf_upload_counts_colored(ctx_, C, pk.data(), pc.data(), n.data(), mn.data(), mx.data(), both.data());
        for (uint32_t c = 0; c < C; ++c) { pf_device_free(ctx_, pk[c]); pf_device_free(ctx_, pc[c]); }
        if (st != PF_OK) { fail(st, std::string("CCDBG::CCDBG():Error: ") + pf_last_error(ctx_)); return; }
        // the colour sets the calling phase asks about go to the device as well (pf_call_set_colours): the mask of colours on every
        // k-mer, UnitigColors::size(), and one bit per k-mer for a colour on part of a unitig
        if (const char *e = getenv("PF_CALL")) resident_ = strcmp(e, "host") != 0;
        if (resident_) {
            const ColorSets &cs = cg_.colors;
            const uint32_t N = g_.n();
            std::vector<uint32_t> us;
            us.reserve(cs.partial.size());
            for (const auto &kv : cs.partial) us.push_back(kv.first);
            std::sort(us.begin(), us.end());
            std::vector<uint32_t> part_first((size_t)N + 1, 0), part_colour;
            std::vector<uint64_t> part_word, bits;
            for (uint32_t u : us) part_first[(size_t)u + 1] = (uint32_t)cs.partial.at(u).size();
            for (uint32_t u = 0; u < N; ++u) part_first[(size_t)u + 1] += part_first[u];
            for (uint32_t u : us)
                for (const ColorSets::Partial &pt : cs.partial.at(u)) {
                    part_colour.push_back(pt.colour);
                    part_word.push_back(bits.size());
                    bits.insert(bits.end(), pt.bits.begin(), pt.bits.end());
                }
            st = pf_call_set_colours(ctx_, C, cs.full_mask.data(), cs.size_total.data(), part_first.data(), part_colour.data(), part_word.data(), bits.data(),
                                     part_colour.size(), bits.size());
            colored_resident_ = st == PF_OK;
        }
    }
    if (!quiet_) printf("CCDBG::CCDBG():CCDBG initialized!\n");
}

int CCDBG::ploidyEstimation_multithread_ptr(const std::string &outpre, const std::vector<std::pair<int, int>> &cutoff, const size_t &thr) {
    return ploidy_estimation(outpre, cutoff, thr);
}

}  // namespace pfh
