// The order-dependent commits of extractSuperBubble_ptr (reference src/CDBG.cpp:373-413, 552-846; colored twin
// src/CCDBG.cpp:2349-2660) over MyUnitig state (src/MyUnitig.hpp:5-136), written once over an accessor so that the same text
// serves the sequential replay (one flag byte per unitig), the parallel replay (one flag byte per *side*, so that two threads
// owning the two sides of one unitig never touch the same byte) and the footprint check of the component model.
//
// What one record touches (the model pf_side_components and SideComponents build on):
//   * its entrance's side and its exit's side: partner slot + that side's LINK / STRICT / COMPLEX bits;
//   * both sides of every other list entry (poison), NON_SUPER of those unitigs;
//   * through release(): the side of a partner that points back -- a link only ever joins two sides that one accepted record
//     touched together -- or, when that side has been re-linked to another unitig since, the partner's minus side
//     ("if (ex->get_plus() == me) set_plus_self(); else set_minus_self();"): unitigs with a side that accepted records link to
//     two different sides (of two unitigs, or both sides of one) therefore count as one unit (both sides joined), and so does
//     the exit of a rejected traversal, whose side is released even when it holds a self-mark;
//   * NON_SUPER of an endpoint is read: it is only ever written by a record that has the unitig as an interior vertex, which
//     joins both of its sides to that record's component.
#pragma once
#include <cstdint>
#include <vector>

#include "ploidyfrost_hip.h"

// the commits compile for the host (g++) and, inside pf_replay.hip, for the device
#if defined(__HIPCC__)
#define PF_HD __host__ __device__
#else
#define PF_HD
#endif

namespace pfh {

namespace state_bits {
constexpr uint8_t B_PLUS = 0x01, B_MINUS = 0x02, B_NON_SUPER = 0x04, B_STRICT_M = 0x08, B_STRICT_P = 0x10, B_COMPLEX_M = 0x20,
                  B_COMPLEX_P = 0x40;
// per-side byte of the parallel replay
constexpr uint8_t S_LINK = 0x01, S_STRICT = 0x02, S_COMPLEX = 0x04, S_NON_SUPER = 0x08;
}  // namespace state_bits

PF_HD inline bool plus_side_of(uint32_t ov) { return (ov & 1) == 0; }

// one flag byte per unitig, plain accesses: the sequential replay
struct FlagsPerUnitig {
    uint8_t *f;
    uint32_t *plus, *minus;
    uint32_t link(uint32_t u, bool ps) const { return ps ? plus[u] : minus[u]; }
    bool plus_points_to(uint32_t ex, uint32_t me) const { return plus[ex] == me + 1; }
    void set_link(uint32_t u, bool ps, uint32_t v, bool real) {
        (ps ? plus[u] : minus[u]) = v;
        const uint8_t b = ps ? state_bits::B_PLUS : state_bits::B_MINUS;
        if (real) f[u] |= b; else f[u] &= (uint8_t)~b;
    }
    void mark_strict(uint32_t u, bool ps) { f[u] |= ps ? state_bits::B_STRICT_P : state_bits::B_STRICT_M; }
    void mark_complex(uint32_t u, bool ps) { f[u] |= ps ? state_bits::B_COMPLEX_P : state_bits::B_COMPLEX_M; }
    bool non_super(uint32_t u, bool) const { return (f[u] & state_bits::B_NON_SUPER) != 0; }
    void set_non_super(uint32_t u) { f[u] |= state_bits::B_NON_SUPER; }
    void begin_record(const pf_bfs_record &) {}
};

// one flag byte per side; partner slots through relaxed atomics (release() looks at the plus slot of a unitig whose plus side
// may belong to another thread's component: the comparison cannot come out true then, whatever that thread is writing)
struct FlagsPerSide {
    uint8_t *f2;   // [2u] plus side, [2u + 1] minus side
    uint32_t *plus, *minus;
    uint32_t link(uint32_t u, bool ps) const { return __atomic_load_n(ps ? &plus[u] : &minus[u], __ATOMIC_RELAXED); }
    bool plus_points_to(uint32_t ex, uint32_t me) const { return __atomic_load_n(&plus[ex], __ATOMIC_RELAXED) == me + 1; }
    void set_link(uint32_t u, bool ps, uint32_t v, bool real) {
        __atomic_store_n(ps ? &plus[u] : &minus[u], v, __ATOMIC_RELAXED);
        uint8_t &b = f2[2 * (size_t)u + (ps ? 0 : 1)];
        if (real) b |= state_bits::S_LINK; else b &= (uint8_t)~state_bits::S_LINK;
    }
    void mark_strict(uint32_t u, bool ps) { f2[2 * (size_t)u + (ps ? 0 : 1)] |= state_bits::S_STRICT; }
    void mark_complex(uint32_t u, bool ps) { f2[2 * (size_t)u + (ps ? 0 : 1)] |= state_bits::S_COMPLEX; }
    bool non_super(uint32_t u, bool ps) const { return (f2[2 * (size_t)u + (ps ? 0 : 1)] & state_bits::S_NON_SUPER) != 0; }
    void set_non_super(uint32_t u) {   // (the caller owns both sides)
        f2[2 * (size_t)u] |= state_bits::S_NON_SUPER;
        f2[2 * (size_t)u + 1] |= state_bits::S_NON_SUPER;
    }
    void begin_record(const pf_bfs_record &) {}
    PF_HD static uint8_t merged(uint8_t p, uint8_t m) {
        using namespace state_bits;
        return (uint8_t)(((p & S_LINK) ? B_PLUS : 0) | ((m & S_LINK) ? B_MINUS : 0) | (((p | m) & S_NON_SUPER) ? B_NON_SUPER : 0) |
                         ((p & S_STRICT) ? B_STRICT_P : 0) | ((m & S_STRICT) ? B_STRICT_M : 0) | ((p & S_COMPLEX) ? B_COMPLEX_P : 0) |
                         ((m & S_COMPLEX) ? B_COMPLEX_M : 0));
    }
};

// the same with every side written noted down: the caller ships exactly those sides somewhere else (pf_replay_finish)
struct FlagsPerSideLogged {
    FlagsPerSide base;
    std::vector<uint32_t> *written;
    uint32_t link(uint32_t u, bool ps) const { return base.link(u, ps); }
    bool plus_points_to(uint32_t ex, uint32_t me) const { return base.plus_points_to(ex, me); }
    void set_link(uint32_t u, bool ps, uint32_t v, bool real) { written->push_back(2 * u + (ps ? 0 : 1)); base.set_link(u, ps, v, real); }
    void mark_strict(uint32_t u, bool ps) { written->push_back(2 * u + (ps ? 0 : 1)); base.mark_strict(u, ps); }
    void mark_complex(uint32_t u, bool ps) { written->push_back(2 * u + (ps ? 0 : 1)); base.mark_complex(u, ps); }
    bool non_super(uint32_t u, bool ps) const { return base.non_super(u, ps); }
    void set_non_super(uint32_t u) { written->push_back(2 * u); written->push_back(2 * u + 1); base.set_non_super(u); }
    void begin_record(const pf_bfs_record &) {}
};

// Hooks of the colored path (src/CCDBG.cpp:2351-2384, 2530-2621); the single-sample replay passes NoColours.
struct NoColours {
    static constexpr bool colored = false;
    template <class A> PF_HD bool allow(A &, const pf_bfs_record &, const uint32_t *) const { return true; }
};

// The colored accept commit's extra gates (src/CCDBG.cpp:2530-2621) over plain arrays (host vectors or device buffers): both
// endpoints carry every colour on every k-mer, and every colour a vertex of the bubble carries in full continues, in full, on
// one of its successors.  An endpoint whose colour set is incomplete is marked NON_SUPER -- a write to the whole unitig by a
// record that owns one side of it: the component model joins both sides of such endpoints (incomplete_entrance / _exit).
struct ColourGate {
    static constexpr bool colored = true;
    uint32_t n_colors = 0;
    int k = 0;
    const uint32_t *len_bp = nullptr;       // unitig lengths
    uint32_t words = 1;                     // 64-bit words per unitig in full_mask: (n_colors + 63) / 64
    const uint64_t *full_mask = nullptr;    // [u * words + c / 64] bit c % 64: colour c on every k-mer of the unitig
    const uint64_t *size_total = nullptr;   // UnitigColors::size(um) with the unitig's own mapping
    const uint32_t *n_full_enc = nullptr;   // colours the file's pair encoding stores as "full"
    const uint32_t *succ = nullptr;         // CSR rows [2N][4]

    PF_HD uint64_t km(uint32_t u) const { return (uint64_t)len_bp[u] - (uint32_t)k + 1; }
    // UnitigColors::size(um) of unitig u's set evaluated with a mapping of km_of k-mers (only the pair encoding looks at the
    // mapping, ColorSet.cpp:902-907; CCDBG.cpp:2552 passes the ENTRANCE's mapping to the exit's set)
    PF_HD uint64_t size_with(uint32_t u, uint64_t km_own, uint64_t km_of) const {
        return size_total[u] - (uint64_t)n_full_enc[u] * km_own + (uint64_t)n_full_enc[u] * km_of;
    }
    PF_HD bool incomplete_entrance(uint32_t su) const { const uint64_t m = km(su); return size_with(su, m, m) != m * n_colors; }
    PF_HD bool incomplete_exit(uint32_t tu, uint32_t su) const { const uint64_t m = km(tu); return size_with(tu, m, km(su)) != m * n_colors; }

    template <class A>
    PF_HD bool allow(A &a, const pf_bfs_record &r, const uint32_t *list) const {
        const uint32_t s = r.entrance, su = s >> 1, t = r.exit, tu = t >> 1;
        bool f = true;
        if (incomplete_entrance(su)) { f = false; a.set_non_super(su); }
        if (incomplete_exit(tu, su)) { f = false; a.set_non_super(tu); }
        if (!f) return false;
        for (uint32_t i = 0; i < r.n_list; ++i) {
            const uint32_t w = list[i];
            if (w == t) continue;
            const uint32_t *row = &succ[(size_t)w * 4];
            const bool endpoint = (w >> 1) == su || (w >> 1) == tu;
            for (uint32_t x = 0; x < words; ++x) {
                // the reference keys its colour lists by unitig id and pre-loads both endpoints with every colour
                const uint32_t left = n_colors - 64 * x;   // colours of this word
                const uint64_t all = left >= 64 ? ~0ull : ((1ull << left) - 1);
                const uint64_t mine = endpoint ? all : full_mask[(size_t)(w >> 1) * words + x];
                uint64_t cont = 0;
                for (int b = 0; b < 4; ++b)
                    if (row[b] != 0xFFFFFFFFu) cont |= full_mask[(size_t)(row[b] >> 1) * words + x];
                if ((cont & mine) != mine) return false;
            }
        }
        return true;
    }
};

template <class Acc, class Col = NoColours>
struct Commits {
    Acc a;
    size_t complex_size;
    Col col;

    PF_HD void side_self(uint32_t u, bool ps) { a.set_link(u, ps, u + 1, false); }
    // "if (ex->get_plus() == me) ex->set_plus_self(); else ex->set_minus_self();"
    PF_HD void release(uint32_t ex, uint32_t me) { side_self(ex, a.plus_points_to(ex, me)); }
    // interior vertex of any committed traversal (e.g. src/CDBG.cpp:800-826)
    PF_HD void poison(uint32_t u) {
        uint32_t p = a.link(u, true);
        if (p != 0 && p != u + 1) release(p - 1, u);
        side_self(u, true);
        p = a.link(u, false);
        if (p != 0 && p != u + 1) release(p - 1, u);
        side_self(u, false);
        a.set_non_super(u);
    }
    PF_HD bool gate_open(uint32_t entrance_ov) const { return a.link(entrance_ov >> 1, plus_side_of(entrance_ov)) == 0; }

    // Order-dependent part of extractSuperBubble_ptr: the three setNoBubble commits (src/CDBG.cpp:552-846) and the no-exit
    // tail (:373-413), applied to one traversal record; the caller has applied the `partner == NULL` gate (:206, 211).
    PF_HD void replay(const pf_bfs_record &r, const uint32_t *list) {
        a.begin_record(r);
        const uint32_t s = r.entrance, su = s >> 1;
        if (r.outcome == PF_BFS_NONE) {
            if (!r.flag_cycle) return;
            for (uint32_t i = 0; i < r.n_list; ++i) poison(list[i] >> 1);
            side_self(su, plus_side_of(s));
            return;
        }
        const uint32_t t = r.exit, tu = t >> 1;
        if (r.outcome == PF_BFS_CYCLE_EXIT) {  // setNoBubble_ptr_cycle
            if (Col::colored) {
                // src/CCDBG.cpp:2351-2384: a side is self-marked only if it held a real partner
                for (uint32_t i = 0; i < r.n_list; ++i) {
                    const uint32_t w = list[i] >> 1;
                    uint32_t p = a.link(w, true);
                    if (p != 0 && p != w + 1) { release(p - 1, w); side_self(w, true); }
                    p = a.link(w, false);
                    if (p != 0 && p != w + 1) { release(p - 1, w); side_self(w, false); }
                    a.set_non_super(w);
                }
            } else
                for (uint32_t i = 0; i < r.n_list; ++i) poison(list[i] >> 1);
            side_self(su, plus_side_of(s));
            side_self(tu, !plus_side_of(t));
            return;
        }
        if (r.outcome == PF_BFS_REJECT) {  // setNoBubble_ptr(seen, p)
            uint32_t p = a.link(su, plus_side_of(s));
            if (p != 0) release(p - 1, su);
            side_self(su, plus_side_of(s));
            p = a.link(tu, !plus_side_of(t));
            if (p != 0) release(p - 1, tu);
            side_self(tu, !plus_side_of(t));
            for (uint32_t i = 0; i < r.n_list; ++i)
                if (list[i] != s && list[i] != t) poison(list[i] >> 1);
            return;
        }
        // PF_BFS_ACCEPT: setNoBubble_ptr(p, seen)
        if (r.n_seen < 4) return;
        if (a.non_super(tu, !plus_side_of(t)) || a.non_super(su, plus_side_of(s))) {
            for (uint32_t i = 0; i < r.n_list; ++i) {
                const uint32_t w = list[i];
                if (w == s) side_self(su, plus_side_of(s));
                else if (w == t) side_self(tu, !plus_side_of(t));
                else poison(w >> 1);
            }
            return;
        }
        if (r.strict) {  // n_seen <= 6 and the structural test, evaluated on the device
            a.mark_strict(su, plus_side_of(s));
            a.mark_strict(tu, !plus_side_of(t));
        }
        if (r.n_seen > complex_size) {
            a.mark_complex(su, plus_side_of(s));
            a.mark_complex(tu, !plus_side_of(t));
        }
        for (uint32_t i = 0; i < r.n_list; ++i)
            if (list[i] != s && list[i] != t) poison(list[i] >> 1);
        if (Col::colored && !col.allow(a, r, list)) {
            side_self(su, plus_side_of(s));
            side_self(tu, !plus_side_of(t));
            return;
        }
        a.set_link(su, plus_side_of(s), tu + 1, true);
        a.set_link(tu, !plus_side_of(t), su + 1, true);
    }
};

// sides: 2u = plus side of unitig u, 2u + 1 = its minus side.  A traversal leaves its entrance s = 2u + strand through side s
// and enters its exit t through side t ^ 1.
PF_HD inline uint32_t entrance_side(uint32_t s) { return s; }
PF_HD inline uint32_t exit_side(uint32_t t) { return t ^ 1u; }

// true when a record can change state at all (whatever the gate says): what the component model has to look at
PF_HD inline bool record_effective(const pf_bfs_record &r) {
    if (r.outcome == PF_BFS_NONE) return r.flag_cycle != 0;
    if (r.outcome == PF_BFS_ACCEPT) return r.n_seen >= 4;
    return true;
}

}  // namespace pfh
