// layout_sim — CPU model of the cache lines a ray batch touches under a device tree layout.
//
// Development tool (not product, not oracle): builds the k-d tree with the product's host builder,
// walks a sample of rays through the HOST tree recording which nodes, id-list entries and triangles
// each ray reads, and then prices those reads under different device layouts (kd_layout.h):
// for every layout, the 128-byte lines are ranked by how many rays touch them, the most-touched
// `hot` lines are taken to live in the 4 MB L2 of an XCD, and the report is the number of DISTINCT
// non-hot lines per ray, by region.  On the round-1 layout the GPU counter (TCC_MISS) says 17.9
// lines per ray; the model is calibrated against that.
//
// build: make -C tools/layout_sim        run: tools/layout_sim/layout_sim /tmp/layout_sim [hot_lines]
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../gigl_amd/csrc/kd_build.h"
#include "../../gigl_amd/csrc/kd_layout.h"

using gigl::KdNode;

static std::vector<char> slurp(const std::string& p) {
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(1); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<char> b((size_t)n);
    if (fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", p.c_str()); exit(1); }
    fclose(f);
    return b;
}

// one read of a ray: kind 0 node (host index), 1 id-list entry (host leaf index, position), 2 triangle (id)
struct Ev { uint32_t kind : 2; uint32_t pos : 30; uint32_t idx; };

static float eps_of(float t) {
    int e;
    (void)frexpf(t, &e);
    return ldexpf(FLT_EPSILON, e + 1);
}

static uint64_t g_push_depth[64];   // pushes by stack depth before the push
// leaf visits: [arrived by descent / by pop][odd / even depth][empty / with prims]
static uint64_t g_leaf_visit[2][2][2];
static uint64_t g_near_empty_both[2];   // "both" steps whose near child is an empty leaf, by depth parity of the child

struct Tracer {
    const gigl::KdTreeHost& T;
    const float* V;
    const uint32_t* I;
    std::vector<Ev>* ev;
    bool tri(uint32_t id, const float* o, const float* d, float tmin, float tmax, float& best) const {
        const float* p0 = V + 3 * (size_t)I[4 * (size_t)id];
        const float* p1 = V + 3 * (size_t)I[4 * (size_t)id + 1];
        const float* p2 = V + 3 * (size_t)I[4 * (size_t)id + 2];
        const float e0[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
        const float e1[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
        const float pv[3] = {d[1] * e1[2] - e1[1] * d[2], d[2] * e1[0] - e1[2] * d[0], d[0] * e1[1] - e1[0] * d[1]};
        const float det = e0[0] * pv[0] + e0[1] * pv[1] + e0[2] * pv[2];
        if (det == 0.0f) return false;
        const float inv = 1.0f / det;
        const float tv[3] = {o[0] - p0[0], o[1] - p0[1], o[2] - p0[2]};
        const float q[3] = {tv[1] * e0[2] - e0[1] * tv[2], tv[2] * e0[0] - e0[2] * tv[0], tv[0] * e0[1] - e0[0] * tv[1]};
        const float u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) * inv;
        const float v = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
        if (!(u >= -1e-4f && v >= -1e-4f && u + v <= 1.0f + 2e-4f)) return false;
        const float t = (e1[0] * q[0] + e1[1] * q[1] + e1[2] * q[2]) * inv;
        if (tmin < t && t < tmax && t < best) { best = t; return true; }
        return false;
    }
    void trace(const float* o4, const float* d4) const {
        const float o[3] = {o4[0], o4[1], o4[2]}, d[3] = {d4[0], d4[1], d4[2]};
        const float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
        float tmin = o4[3], tmax = d4[3];
        float entr = -FLT_MAX, exit = FLT_MAX;
        for (int a = 0; a < 3; ++a) {
            float t0 = (T.box_min[a] - o[a]) * inv[a], t1 = (T.box_max[a] - o[a]) * inv[a];
            if (t0 > t1) std::swap(t0, t1);
            if (t0 > entr) entr = t0;
            if (t1 < exit) exit = t1;
        }
        if (!(entr <= exit && entr < tmax && exit > tmin)) return;
        tmin = std::max(tmin, entr - eps_of(entr));
        tmax = std::min(tmax, exit + eps_of(exit));
        struct St { float tmin, tmax; uint32_t id; int depth; } stack[64];
        int depth = 0, by_pop = 0;
        int sp = 0;
        uint32_t node = 0;
        float best = FLT_MAX;
        const std::vector<KdNode>& H = T.nodes;
        for (;;) {
            if (best < tmin) break;
            ev->push_back(Ev{0u, 0u, node});
            while (!(H[node].w1 & 1u)) {
                const unsigned a = (H[node].w1 >> 1) & 3u;
                float split;
                memcpy(&split, &H[node].w0, 4);
                const bool bottom_first = o[a] < split || (o[a] == split && d[a] <= 0.0f);
                const uint32_t below = node + 1u, above = H[node].w1 >> 3;
                const uint32_t first = bottom_first ? below : above, second = bottom_first ? above : below;
                const float ts = (split - o[a]) * inv[a];
                if (ts > tmax || ts <= 0.0f) node = first;
                else if (ts < tmin) node = second;
                else {
                    ++g_push_depth[sp];
                    stack[sp++] = St{ts - eps_of(ts), tmax, second, depth + 1};
                    node = first;
                    tmax = ts + eps_of(ts);
                    if ((H[node].w1 & 1u) && (H[node].w1 >> 1) == 0u) ++g_near_empty_both[(depth + 1) & 1];
                }
                ++depth;
                by_pop = 0;
                ev->push_back(Ev{0u, 0u, node});
            }
            const uint32_t n = H[node].w1 >> 1;
            ++g_leaf_visit[by_pop][depth & 1][n ? 1 : 0];
            bool hit = false;
            for (uint32_t i = 0; i < n; ++i) {
                uint32_t id = H[node].w0;
                if (n >= 2) {
                    id = T.leaf_prim_ids[H[node].w0 + i];
                    ev->push_back(Ev{1u, i, node});
                }
                ev->push_back(Ev{2u, 0u, id});
                hit = tri(id, o, d, tmin, tmax, best) || hit;
            }
            if (hit) break;
            if (sp == 0) break;
            --sp;
            node = stack[sp].id; tmin = stack[sp].tmin; tmax = stack[sp].tmax; depth = stack[sp].depth; by_pop = 1;
        }
    }
};

struct TriLayout {
    std::vector<uint32_t> pos;   // triangle id -> record index
    uint32_t rec_bytes = 64;     // record size
    uint32_t per_line = 2;       // records per 128-byte line (records never straddle a line)
    uint64_t addr(uint32_t id) const {
        const uint32_t p = pos.empty() ? id : pos[id];
        return (uint64_t)(p / per_line) * 128u + (uint64_t)(p % per_line) * rec_bytes;
    }
};

struct Report { double node, ids, tri, all, node_all, ids_all, tri_all; size_t bytes; };

// abstract layouts: only a line number per host node (id lists are taken to share the leaf's line)
static const std::vector<uint32_t>* g_abstract_lines = nullptr;
static const std::vector<uint32_t>* g_list_dw = nullptr;   // triple layout: dword address of a leaf's extra record
static size_t g_abstract_bytes = 0;

static Report price(const gigl::KdTreeHost& T, const gigl::DeviceTree& D, const std::vector<uint32_t>& slot_of,
                    const TriLayout& TL, const std::vector<Ev>& ev, const std::vector<size_t>& ray_begin,
                    size_t hot_lines) {
    // line keys: region << 40 | line
    auto line_of = [&](const Ev& e) -> uint64_t {
        if (g_abstract_lines && e.kind != 2u) return (*g_abstract_lines)[e.idx];
        if (e.kind == 0u) return ((uint64_t)slot_of[e.idx] * 8u) >> 7;
        if (e.kind == 1u) {
            if (D.triples) {
                const uint32_t dw = (*g_list_dw)[e.idx];
                return dw == ~0u ? (((uint64_t)slot_of[e.idx] * 8u) >> 7) : (((uint64_t)dw * 4u) >> 7);
            }
            if (D.ids_inline) {
                if ((T.nodes[e.idx].w1 >> 1) < 3u) return ((uint64_t)slot_of[e.idx] * 8u) >> 7;
                const uint32_t dw = D.slots[slot_of[e.idx]].w0 + e.pos;
                return ((uint64_t)dw * 4u) >> 7;
            }
            if ((T.nodes[e.idx].w1 >> 1) < 3u) return ((uint64_t)slot_of[e.idx] * 8u) >> 7;   // ids in the node
            return (1ull << 40) | (((uint64_t)(T.nodes[e.idx].w0 + e.pos) * 4u) >> 7);
        }
        return (2ull << 40) | (TL.addr(e.idx) >> 7);
    };
    const size_t n_rays = ray_begin.size() - 1;
    std::unordered_map<uint64_t, uint32_t> freq;
    freq.reserve(ev.size() / 2);
    std::vector<uint64_t> tmp;
    for (size_t r = 0; r < n_rays; ++r) {
        tmp.clear();
        for (size_t i = ray_begin[r]; i < ray_begin[r + 1]; ++i) tmp.push_back(line_of(ev[i]));
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        for (uint64_t l : tmp) ++freq[l];
    }
    std::vector<uint32_t> counts;
    counts.reserve(freq.size());
    for (auto& kv : freq) counts.push_back(kv.second);
    uint32_t thr = 0;   // lines touched by more than thr rays are hot
    if (counts.size() > hot_lines) {
        std::nth_element(counts.begin(), counts.begin() + (long)hot_lines, counts.end(), std::greater<uint32_t>());
        thr = counts[hot_lines];
    } else thr = 0xFFFFFFFFu;
    double cold[3] = {0, 0, 0}, all[3] = {0, 0, 0};
    for (auto& kv : freq) {
        const int reg = (int)(kv.first >> 40);
        all[reg] += kv.second;
        if (counts.size() <= hot_lines || kv.second > thr) continue;
        cold[reg] += kv.second;
    }
    Report R;
    const double nr = (double)n_rays;
    R.node = cold[0] / nr; R.ids = cold[1] / nr; R.tri = cold[2] / nr;
    R.all = R.node + R.ids + R.tri;
    R.node_all = all[0] / nr; R.ids_all = all[1] / nr; R.tri_all = all[2] / nr;
    R.bytes = g_abstract_lines ? g_abstract_bytes : D.slots.size() * 8 + (D.ids_inline ? 0 : T.leaf_prim_ids.size() * 4);
    return R;
}

static void show(const char* name, const Report& R) {
    printf("%-34s cold lines/ray: node %6.2f  ids %5.2f  tri %5.2f  total %6.2f | distinct: node %6.2f ids %5.2f tri %5.2f | tree %7.1f MB\n",
           name, R.node, R.ids, R.tri, R.all, R.node_all, R.ids_all, R.tri_all, R.bytes / 1e6);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp/layout_sim";
    const size_t hot = argc > 2 ? (size_t)atol(argv[2]) : 32768;
    auto hdr = slurp(dir + "/header.bin");
    const uint32_t* h = (const uint32_t*)hdr.data();
    const uint32_t nv = h[0], nt = h[1], nr = h[2];
    auto vb = slurp(dir + "/verts.bin"), tb = slurp(dir + "/tris.bin"), ob = slurp(dir + "/ro.bin"), db = slurp(dir + "/rd.bin");
    const float* V = (const float*)vb.data();
    const uint32_t* I = (const uint32_t*)tb.data();
    gigl::KdTreeHost T;
    gigl::KdBuildParams bp;
    const char* err = nullptr;
    if (!gigl::kd_build(V, nv, I, nt, bp, T, &err)) { fprintf(stderr, "build: %s\n", err ? err : "?"); return 1; }
    printf("tree: %zu nodes, %zu leaf ids, depth %d\n", T.nodes.size(), T.leaf_prim_ids.size(), T.actual_depth);
    std::vector<Ev> ev;
    std::vector<size_t> ray_begin(1, 0);
    ev.reserve((size_t)nr * 90);
    Tracer tr{T, V, I, &ev};
    double nodes = 0, tests = 0;
    for (uint32_t r = 0; r < nr; ++r) {
        tr.trace((const float*)ob.data() + 4 * (size_t)r, (const float*)db.data() + 4 * (size_t)r);
        ray_begin.push_back(ev.size());
    }
    for (const Ev& e : ev) { nodes += e.kind == 0u; tests += e.kind == 2u; }
    printf("rays %u: nodes/ray %.2f, tests/ray %.2f, hot set %zu lines\n", nr, nodes / nr, tests / nr, hot);
    {
        uint64_t tot = 0, acc = 0;
        for (int i = 0; i < 64; ++i) tot += g_push_depth[i];
        printf("pushes/ray %.2f; share of pushes landing at stack index >= k:", (double)tot / nr);
        for (int k = 0; k < 20; ++k) {
            if (k >= 6 && k <= 16) printf(" k=%d %.2f%%", k, 100.0 * (double)(tot - acc) / (double)tot);
            acc += g_push_depth[k];
        }
        printf("\n");
        for (int bp = 0; bp < 2; ++bp)
            for (int par = 0; par < 2; ++par)
                printf("leaf visits/ray arrived by %s at %s depth: empty %.2f, with prims %.2f\n", bp ? "pop    " : "descent",
                       par ? "odd " : "even", (double)g_leaf_visit[bp][par][0] / nr, (double)g_leaf_visit[bp][par][1] / nr);
        printf("both-steps/ray whose near child is an empty leaf: odd depth %.2f, even depth %.2f\n",
               (double)g_near_empty_both[1] / nr, (double)g_near_empty_both[0] / nr);
    }

    TriLayout tl64;   // 64-byte records in id order (round 1)
    std::vector<uint32_t> slot_of;
    for (int levels = 3; levels <= 3; ++levels) {
        gigl::DeviceTree D;
        D.slot_of_host = &slot_of;
        gigl::layout_levels(T, levels, D);
        char nm[64];
        snprintf(nm, sizeof nm, "levels=%d, tris 64B id order", levels);
        show(nm, price(T, D, slot_of, tl64, ev, ray_begin, hot));
    }
    // triangle order of first reference in a depth-first walk of the leaves
    TriLayout tl_dfs = tl64;
    {
        tl_dfs.pos.assign(nt, 0xFFFFFFFFu);
        uint32_t next = 0;
        for (const KdNode& n : T.nodes) {
            if (!(n.w1 & 1u)) continue;
            const uint32_t c = n.w1 >> 1;
            for (uint32_t i = 0; i < c; ++i) {
                const uint32_t id = c == 1 ? n.w0 : T.leaf_prim_ids[n.w0 + i];
                if (tl_dfs.pos[id] == 0xFFFFFFFFu) tl_dfs.pos[id] = next++;
            }
        }
        for (uint32_t i = 0; i < nt; ++i) if (tl_dfs.pos[i] == 0xFFFFFFFFu) tl_dfs.pos[i] = next++;
    }
    TriLayout tl40 = tl_dfs;
    tl40.rec_bytes = 40; tl40.per_line = 3;
    {
        gigl::DeviceTree D;
        D.slot_of_host = &slot_of;
        gigl::layout_levels(T, 3, D);
        show("levels=3, tris 64B leaf order", price(T, D, slot_of, tl_dfs, ev, ray_begin, hot));
        show("levels=3, tris 40B leaf order", price(T, D, slot_of, tl40, ev, ray_begin, hot));
    }
    {
        gigl::DeviceTree D;
        D.slot_of_host = &slot_of;
        gigl::layout_clustered(T, D);
        show("clustered, tris 64B id order", price(T, D, slot_of, tl64, ev, ray_begin, hot));
        TriLayout tl40id = tl64;
        tl40id.rec_bytes = 40; tl40id.per_line = 3;
        show("clustered, tris 40B id order", price(T, D, slot_of, tl40id, ev, ray_begin, hot));
        TriLayout tl32id = tl64;   // hypothetical 32-byte records
        tl32id.rec_bytes = 32; tl32id.per_line = 4;
        show("clustered, tris 32B id order", price(T, D, slot_of, tl32id, ev, ray_begin, hot));
    }
    {
        gigl::DeviceTree D;
        std::vector<uint32_t> list_dw;
        D.slot_of_host = &slot_of;
        D.list_dw_of_host = &list_dw;
        const bool fits = gigl::layout_triples(T, D);
        const char* what = nullptr;
        const bool same = gigl::layout_verify(T, D, &what);
        printf("triples: fits %d, verify %d %s\n", (int)fits, (int)same, what ? what : "");
        g_list_dw = &list_dw;
        show("triples, tris 64B id order", price(T, D, slot_of, tl64, ev, ray_begin, hot));
    }
    if (argc > 3) return 0;
    // ---- abstract layouts: what would denser node encodings buy? ----
    const std::vector<KdNode>& H = T.nodes;
    std::vector<uint32_t> lines(H.size(), 0u);
    gigl::DeviceTree none;
    // (A) implicit complete treelets of L levels of pairs, one line each
    for (int L = 3; L <= 5; ++L) {
        uint32_t next_line = 1;
        std::vector<uint32_t> roots(1, 0u);   // interior nodes whose child pair starts a treelet
        lines[0] = 0;
        while (!roots.empty()) {
            const uint32_t r = roots.back();
            roots.pop_back();
            const uint32_t ln = next_line++;
            std::vector<uint32_t> cur(1, r), nxt;
            for (int lv = 0; lv < L; ++lv) {
                nxt.clear();
                for (uint32_t it : cur) {
                    const uint32_t kids[2] = {it + 1u, H[it].w1 >> 3};
                    for (uint32_t k : kids) {
                        lines[k] = ln;
                        if (!(H[k].w1 & 1u)) (lv + 1 < L ? nxt : roots).push_back(k);
                    }
                }
                cur.swap(nxt);
            }
        }
        g_abstract_lines = &lines;
        g_abstract_bytes = (size_t)next_line * 128;
        char nm[64];
        snprintf(nm, sizeof nm, "implicit %d-level treelet per line", L);
        show(nm, price(T, none, slot_of, tl64, ev, ray_begin, hot));
    }
    // (C) greedy clusters with a byte budget per 128-byte line: `pair_bits` per child pair, `hdr_bits` per
    // line, and the id lists of leaves with at least `list_min` prims stored in the leaf's line
    struct Enc { const char* name; uint32_t pair_bits, hdr_bits, list_min; };
    const Enc encs[] = {{"16 B pairs, lists n>=3 (clustered)", 128, 0, 3},
                        {"12 B pairs, lists n>=3", 96, 0, 3},
                        {"12 B pairs, lists n>=2", 96, 0, 2},
                        {"8.75 B pairs + 4 B hdr, lists n>=3", 70, 32, 3},
                        {"8.75 B pairs + 4 B hdr, lists n>=2", 70, 32, 2},
                        {"8.75 B pairs + 4 B hdr, no lists", 70, 32, 1000},
                        {"16 B pairs, no lists", 128, 0, 1000}};
    for (const Enc& E : encs) {
        const uint32_t cap = 1024;
        auto cost_of = [&](uint32_t it) {
            uint32_t c = E.pair_bits;
            const uint32_t kids[2] = {it + 1u, H[it].w1 >> 3};
            for (uint32_t k : kids)
                if ((H[k].w1 & 1u) && (H[k].w1 >> 1) >= E.list_min && (H[k].w1 >> 1) <= 24) c += 32u * (H[k].w1 >> 1);
            return c > cap - E.hdr_bits ? cap - E.hdr_bits : c;
        };
        std::vector<uint32_t> subp(H.size(), 0u);
        for (size_t i = H.size(); i-- > 0;)
            if (!(H[i].w1 & 1u)) {
                const uint64_t v = (uint64_t)cost_of((uint32_t)i) + subp[i + 1] + subp[H[i].w1 >> 3];
                subp[i] = v > 0x7FFFFFFF ? 0x7FFFFFFFu : (uint32_t)v;
            }
        uint32_t next_line = 1, cur_line = 0, free_b = cap - E.hdr_bits - 64;
        std::vector<uint32_t> roots(1, 0u);
        lines[0] = 0;
        const uint32_t room = cap - E.hdr_bits;
        while (!roots.empty()) {
            const uint32_t r = roots.back();
            roots.pop_back();
            if (subp[r] <= room ? subp[r] > free_b : (free_b < room / 2 || cost_of(r) > free_b)) { cur_line = next_line++; free_b = room; }
            std::vector<uint32_t> q(1, r), big, small;
            size_t qi = 0;
            while (qi < q.size()) {
                const uint32_t it = q[qi++];
                const uint32_t c = cost_of(it);
                if (c > free_b) { (subp[it] <= room ? small : big).push_back(it); continue; }
                free_b -= c;
                const uint32_t kids[2] = {it + 1u, H[it].w1 >> 3};
                for (uint32_t k : kids) {
                    lines[k] = cur_line;
                    if (!(H[k].w1 & 1u)) q.push_back(k);
                }
            }
            for (size_t i = big.size(); i-- > 0;) roots.push_back(big[i]);
            for (size_t i = small.size(); i-- > 0;) roots.push_back(small[i]);
        }
        g_abstract_lines = &lines;
        g_abstract_bytes = (size_t)next_line * 128;
        show(E.name, price(T, none, slot_of, tl64, ev, ray_begin, hot));
    }
    return 0;
}
