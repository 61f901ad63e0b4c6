// Constant structure of the structured-coalescent chains, derived on the host at
// library load from the model definition (not from any table of the reference):
//
//   two populations, 2+2 sampled haplotypes: 44 states = multisets of lineages
//   (d0, d1, pop) with sum d0 = sum d1 = 2; index layout as in the reference
//   (TwoPopulations.MapIndToState, TwoPopulations.py:130-186) because
//   CollapsePops (MigrationInference.py:518-528), AncientSampleP0
//   (TwoPopulations.py:246-262) and the initial state e_2 (:469-471) are defined
//   on that layout.
//
// Everything the kernels need is a per-destination-row view of the generator
//   M = la0*A0 + la1*A1 + mu0*B0 + mu1*B1          (UpdateMatrixCol :336-359)
// plus StateToJAF (:188-219), the pulse operator (:361-377) and the 44 -> 8 map.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <map>
#include <stdexcept>
#include <tuple>
#include <vector>

#include "misti_consts.h"

namespace misti {

struct Lineage { int d0, d1, pop; };
using State = std::vector<Lineage>;

inline void canon(State& s) {
    std::stable_sort(s.begin(), s.end(), [](const Lineage& a, const Lineage& b) {
        int wa = a.d0 + a.d1, wb = b.d0 + b.d1;
        if (wa != wb) return wa > wb;
        if (a.d0 != b.d0) return a.d0 > b.d0;
        return a.pop < b.pop;
    });
}
inline std::vector<int> key(const State& s) {
    std::vector<int> k;
    for (auto& l : s) { k.push_back(l.d0); k.push_back(l.d1); k.push_back(l.pop); }
    return k;
}

// index -> state
inline State decode2(int ind) {
    State s;
    auto single = [&](int d0, int d1, int count_in_pop1) {
        for (int k = 0; k < 2; ++k) s.push_back({d0, d1, k < count_in_pop1 ? 1 : 0});
    };
    if (ind < 9) { single(1, 0, ind / 3); single(0, 1, ind % 3); }
    else if (ind < 15) { int r = ind - 9;  s.push_back({2, 0, r / 3}); single(0, 1, r % 3); }
    else if (ind < 23) { int r = ind - 15; s.push_back({1, 1, r / 4}); s.push_back({1, 0, (r % 4) / 2}); s.push_back({0, 1, r % 2}); }
    else if (ind < 29) { int r = ind - 23; s.push_back({0, 2, r / 3}); single(1, 0, r % 3); }
    else if (ind < 33) { int r = ind - 29; s.push_back({2, 1, r / 2}); s.push_back({0, 1, r % 2}); }
    else if (ind < 37) { int r = ind - 33; s.push_back({1, 2, r / 2}); s.push_back({1, 0, r % 2}); }
    else if (ind < 41) { int r = ind - 37; s.push_back({2, 0, r / 2}); s.push_back({0, 2, r % 2}); }
    else { int r = ind - 41; s.push_back({1, 1, r == 2 ? 1 : 0}); s.push_back({1, 1, r >= 1 ? 1 : 0}); }
    canon(s);
    return s;
}

inline int jaf_class(int d0, int d1) {
    // JSFS columns 0100,1100,0001,0101,1101,0011,0111 (MigrationInference.py:170)
    static const int tab[3][3] = {{-1, 2, 5}, {0, 3, 6}, {1, 4, -1}};
    return tab[d0][d1];
}

struct HostTables {
    // gen[kind][dst][src]; kind 0,1 = coalescence in pop 0,1; 2,3 = migration out of pop 0,1
    std::array<std::array<std::array<int, NS2>, NS2>, 4> gen{};
    std::array<std::array<int, 7>, NS2> jaf{};
    std::array<std::array<int, 7>, NS1> jaf1{};
    std::array<std::array<int, NS1>, NS1> gen1{};           // one-population generator / la
    // row view
    int src[MAXNZ][64]{}, kind[MAXNZ][64]{}, mult[MAXNZ][64]{};
    int dcnt[4][64]{};
    int grp[64]{};                                           // 44 -> 8 collapse group of each state
    int grp_lo[NS1]{}, grp_hi[NS1]{};
    int anc_n[2]{}, anc_src[2][8]{}, anc_dst[2]{};
    int pulse_n[2][64]{}, pulse_src[2][64][MAXPULSE]{}, pulse_ab[2][64][MAXPULSE]{};
    int max_nz = 0, max_pulse = 0;
};

inline HostTables build_tables() {
    HostTables t;
    std::vector<State> states;
    std::map<std::vector<int>, int> index;
    for (int i = 0; i < NS2; ++i) {
        states.push_back(decode2(i));
        index[key(states.back())] = i;
    }
    if ((int)index.size() != NS2) throw std::runtime_error("state codec is not a bijection");
    auto find = [&](State s) -> int {
        canon(s);
        auto it = index.find(key(s));
        return it == index.end() ? -1 : it->second;
    };
    for (int src = 0; src < NS2; ++src) {
        const State& st = states[src];
        for (size_t i = 0; i < st.size(); ++i) {
            State mv = st;
            mv[i].pop = 1 - st[i].pop;
            int dst = find(mv);
            if (dst < 0) throw std::runtime_error("migration leaves the state space");
            t.gen[2 + st[i].pop][dst][src] += 1;
            t.gen[2 + st[i].pop][src][src] -= 1;
            for (size_t j = i + 1; j < st.size(); ++j) {
                if (st[j].pop != st[i].pop) continue;
                State co;
                for (size_t k = 0; k < st.size(); ++k) if (k != i && k != j) co.push_back(st[k]);
                co.push_back({st[i].d0 + st[j].d0, st[i].d1 + st[j].d1, st[i].pop});
                if (co.size() >= 2) {            // the last coalescence is absorption
                    int d = find(co);
                    if (d < 0) throw std::runtime_error("coalescence leaves the state space");
                    t.gen[st[i].pop][d][src] += 1;
                }
                t.gen[st[i].pop][src][src] -= 1;
            }
        }
        for (auto& l : st) t.jaf[src][jaf_class(l.d0, l.d1)] += 1;
    }
    // row view: every off-diagonal entry of a row has exactly one kind
    for (int dst = 0; dst < NS2; ++dst) {
        int n = 0;
        for (int src = 0; src < NS2; ++src) {
            if (src == dst) continue;
            for (int k = 0; k < 4; ++k) {
                int c = t.gen[k][dst][src];
                if (!c) continue;
                if (n >= MAXNZ) throw std::runtime_error("generator row wider than MAXNZ");
                t.src[n][dst] = src; t.kind[n][dst] = k; t.mult[n][dst] = c; ++n;
            }
        }
        t.max_nz = std::max(t.max_nz, n);
        for (; n < MAXNZ; ++n) { t.src[n][dst] = dst; t.kind[n][dst] = 0; t.mult[n][dst] = 0; }
        for (int k = 0; k < 4; ++k) t.dcnt[k][dst] = -t.gen[k][dst][dst];
    }
    for (int l = NS2; l < 64; ++l) for (int n = 0; n < MAXNZ; ++n) { t.src[n][l] = l; }
    // 44 -> 8: forget the populations
    {
        std::vector<State> s1;
        std::map<std::vector<int>, int> idx1;
        for (int i = 0; i < NS2; ++i) {
            State f = states[i];
            for (auto& l : f) l.pop = 0;
            canon(f);
            auto k = key(f);
            if (!idx1.count(k)) { int id = (int)idx1.size(); idx1[k] = id; s1.push_back(f); }
            t.grp[i] = idx1[k];
        }
        if ((int)idx1.size() != NS1) throw std::runtime_error("collapse does not give 8 states");
        for (int g = 0; g < NS1; ++g) {
            t.grp_lo[g] = NS2; t.grp_hi[g] = 0;
            for (int i = 0; i < NS2; ++i) if (t.grp[i] == g) { t.grp_lo[g] = std::min(t.grp_lo[g], i); t.grp_hi[g] = std::max(t.grp_hi[g], i + 1); }
            for (int i = t.grp_lo[g]; i < t.grp_hi[g]; ++i) if (t.grp[i] != g) throw std::runtime_error("collapse groups are not contiguous");
            for (auto& l : s1[g]) t.jaf1[g][jaf_class(l.d0, l.d1)] += 1;
        }
        auto find1 = [&](State s) -> int { for (auto& l : s) l.pop = 0; canon(s); auto it = idx1.find(key(s)); return it == idx1.end() ? -1 : it->second; };
        for (int src = 0; src < NS1; ++src) {
            const State& st = s1[src];
            for (size_t i = 0; i < st.size(); ++i) for (size_t j = i + 1; j < st.size(); ++j) {
                State co;
                for (size_t k = 0; k < st.size(); ++k) if (k != i && k != j) co.push_back(st[k]);
                co.push_back({st[i].d0 + st[j].d0, st[i].d1 + st[j].d1, 0});
                if (co.size() >= 2) t.gen1[find1(co)][src] += 1;
                t.gen1[src][src] -= 1;
            }
        }
    }
    // ancient second genome: everything of genome 2 is put back to "not yet sampled"
    for (int i = 0; i < NS2; ++i) {
        int n10 = 0, n20 = 0;
        for (auto& l : states[i]) { n10 += (l.d0 == 1 && l.d1 == 0 && l.pop == 0); n20 += (l.d0 == 2 && l.d1 == 0 && l.pop == 0); }
        if (n10 == 2) t.anc_src[0][t.anc_n[0]++] = i;
        if (n20 == 1) t.anc_src[1][t.anc_n[1]++] = i;
    }
    t.anc_dst[0] = 2; t.anc_dst[1] = 11;
    // pulse: each lineage in population `from` moves independently with probability r
    for (int from = 0; from < 2; ++from) {
        std::map<std::tuple<int, int, int, int>, int> ent;   // (dst, src, stay, move) -> multiplicity
        for (int src = 0; src < NS2; ++src) {
            const State& st = states[src];
            std::vector<int> movers;
            for (size_t k = 0; k < st.size(); ++k) if (st[k].pop == from) movers.push_back((int)k);
            for (int mask = 0; mask < (1 << movers.size()); ++mask) {
                State ns = st;
                int mv = 0;
                for (size_t b = 0; b < movers.size(); ++b) if (mask >> b & 1) { ns[movers[b]].pop = 1 - from; ++mv; }
                ent[{find(ns), src, (int)movers.size() - mv, mv}] += 1;
            }
        }
        for (auto& e : ent) {
            int dst = std::get<0>(e.first), src = std::get<1>(e.first), a = std::get<2>(e.first), b = std::get<3>(e.first);
            int& n = t.pulse_n[from][dst];
            if (n >= MAXPULSE) throw std::runtime_error("pulse row wider than MAXPULSE");
            t.pulse_src[from][dst][n] = src;
            t.pulse_ab[from][dst][n] = (e.second << 8) | (a << 4) | b;   // multiplicity, stay, move
            ++n;
            t.max_pulse = std::max(t.max_pulse, n);
        }
    }
    return t;
}

}  // namespace misti
