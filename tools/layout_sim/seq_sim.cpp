// seq_sim — CPU emulation of the traversal kernel's LOAD SEQUENCE over a device layout.
//
// Development tool (not product, not oracle).  layout_sim counts the distinct lines a ray touches;
// the GPU counters say that is the wrong measure: a line stays in an XCD's L2 for a few microseconds
// only (4 MB against ~6 G missing lines per second per XCD), a ray lives ~130 us, so a line a ray
// comes back to after a pop has usually been evicted.  What the chip pays for is every load whose
// line is neither globally hot (the top levels of the tree, re-touched every few microseconds by
// some ray) nor among the last few lines this ray itself touched.  This tool walks each sample ray
// through the DEVICE layout exactly as trace_kernel does (same loads, same order, pops included) and
// prices the sequence with that model: misses = loads whose line is not in the ray's last K distinct
// lines and not in the hot set (lines touched by more than `hot_frac` of the rays).
//
// build: make -C tools/layout_sim seq_sim     run: tools/layout_sim/seq_sim /tmp/layout_sim [K=2] [hot_frac=5e-4]
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

static float eps_of(float t) {
    int e;
    (void)frexpf(t, &e);
    return ldexpf(FLT_EPSILON, e + 1);
}
static float asf(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

constexpr uint64_t kTriRegion = 1ull << 40;
// load categories (bits 44-47 of a sequence entry; stripped before the line is formed)
enum Cat { C_ROOT = 0, C_DESC = 1, C_POP = 2, C_LEAFREC = 3, C_IDS = 4, C_TRI = 5, C_FARVAL = 6, C_N = 8 };
static const char* kCatName[C_N] = {"root", "descent rec", "pop rec", "leaf rec", "id array", "triangle", "far-by-value", "-"};
static uint64_t tag(uint64_t a, Cat c) { return a | ((uint64_t)c << 44); }

struct Scene {
    const float* V;
    const uint32_t* I;
    float box_min[3], box_max[3];
    uint32_t tri_rec_bytes = 64;   // triangle record stride
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
};

struct Ray { float o[3], d[3], inv[3], tmin, tmax; };

static bool setup(const Scene& S, const float* o4, const float* d4, Ray& r) {
    for (int a = 0; a < 3; ++a) { r.o[a] = o4[a]; r.d[a] = d4[a]; r.inv[a] = 1.0f / d4[a]; }
    r.tmin = o4[3]; r.tmax = d4[3];
    float entr = -FLT_MAX, exit = FLT_MAX;
    for (int a = 0; a < 3; ++a) {
        float t0 = (S.box_min[a] - r.o[a]) * r.inv[a], t1 = (S.box_max[a] - r.o[a]) * r.inv[a];
        if (t0 > t1) std::swap(t0, t1);
        if (t0 > entr) entr = t0;
        if (t1 < exit) exit = t1;
    }
    if (!(entr <= exit && entr < r.tmax && exit > r.tmin)) return false;
    r.tmin = std::max(r.tmin, entr - eps_of(entr));
    r.tmax = std::min(r.tmax, exit + eps_of(exit));
    return true;
}

struct Step { float ts; bool bottom_first, both, take_top; };
static Step decide(const Ray& r, unsigned a, float split) {
    Step s;
    s.bottom_first = r.o[a] < split || (r.o[a] == split && r.d[a] <= 0.0f);
    s.ts = (split - r.o[a]) * r.inv[a];
    const bool only_first = s.ts > r.tmax || s.ts <= 0.0f;
    const bool only_second = s.ts < r.tmin;
    s.both = !(only_first || only_second);
    s.take_top = s.bottom_first == (!only_first && only_second);
    return s;
}

// ---- triple records (kd_layout.h: layout_triples), as trace_kernel<TRIPLES> walks them ----
// by_value: the variant where a lvl-1 "both" step loads BOTH child records and pushes the far one by
// value (no load at the pop)
static void walk_triples(const Scene& S, const uint32_t* W, const Ray& r0, bool by_value, std::vector<uint64_t>& seq,
                         uint64_t* n_loads) {
    Ray r = r0;
    struct E { float tmin, tmax; uint32_t w0, w1; uint32_t rec[4]; bool has_rec; } st[64];
    int sp = 0;
    uint32_t rec[4] = {W[0], W[1], W[2], W[3]};
    seq.push_back(0);
    unsigned lvl = 0;
    float best = FLT_MAX;
    auto load_rec = [&](uint32_t u, uint32_t* out, Cat c) {
        seq.push_back(tag((uint64_t)u * 16u, c));
        ++*n_loads;
        memcpy(out, W + 4 * (size_t)u, 16);
    };
    for (;;) {
        if (best < r.tmin) return;
        // T-phase
        while (((lvl == 0 ? rec[3] : rec[1]) & 3u) != 3u) {
            if (lvl == 0) {
                const Step s = decide(r, rec[3] & 3u, asf(rec[0]));
                const uint32_t ta = (rec[3] >> 2) & 3u, tb = (rec[3] >> 4) & 3u;
                const uint32_t blk = (rec[3] >> 6) << 1, blk_b = blk + (ta != 3u ? 2u : 0u);
                const uint32_t da = ta | (blk << 3), db = tb | (blk_b << 3);
                if (s.both) {
                    E& e = st[sp++];
                    e.tmin = s.ts - eps_of(s.ts); e.tmax = r.tmax;
                    e.w0 = s.bottom_first ? rec[2] : rec[1]; e.w1 = s.bottom_first ? db : da; e.has_rec = false;
                    r.tmax = s.ts + eps_of(s.ts);
                }
                rec[0] = s.take_top ? rec[2] : rec[1];
                rec[1] = s.take_top ? db : da;
                lvl = 1;
            }
            if ((rec[1] & 3u) != 3u) {
                const Step s = decide(r, rec[1] & 3u, asf(rec[0]));
                const uint32_t blk = rec[1] >> 3;
                const uint32_t near_u = blk + (s.take_top ? 1u : 0u);
                if (s.both) {
                    E& e = st[sp++];
                    e.tmin = s.ts - eps_of(s.ts); e.tmax = r.tmax;
                    e.w0 = 0; e.w1 = 4u | ((blk + (s.bottom_first ? 1u : 0u)) << 3); e.has_rec = false;
                    if (by_value) { load_rec(e.w1 >> 3, e.rec, C_FARVAL); e.has_rec = true; }
                    r.tmax = s.ts + eps_of(s.ts);
                }
                load_rec(near_u, rec, C_DESC);
                lvl = 0;
            }
        }
        // L-phase
        if (lvl == 1) {
            const uint32_t payload = rec[0], kind = payload & 3u;
            if (kind == 2u) load_rec(payload >> 2, rec, C_LEAFREC);
            else { rec[0] = payload >> 2; rec[1] = rec[2] = 0; rec[3] = kind == 1u ? (3u | (1u << 2)) : 3u; }
            lvl = 0;
        }
        const uint32_t n = rec[3] >> 2;
        bool hit = false;
        for (uint32_t i = 0; i < n; i += 2) {
            uint32_t id0, id1;
            if (n <= 3u) { id0 = i == 0 ? rec[0] : rec[2]; id1 = rec[1]; }
            else {
                const uint64_t a = (uint64_t)rec[0] * 16u + (uint64_t)(i >> 1) * 8u;
                seq.push_back(tag(a, C_IDS)); ++*n_loads;
                id0 = W[4 * (size_t)rec[0] + i]; id1 = i + 1 < n ? W[4 * (size_t)rec[0] + i + 1] : id0;
            }
            seq.push_back(tag(kTriRegion | (uint64_t)id0 * S.tri_rec_bytes, C_TRI)); ++*n_loads;
            hit = S.tri(id0, r.o, r.d, r.tmin, r.tmax, best) || hit;
            if (i + 1 < n) {
                seq.push_back(tag(kTriRegion | (uint64_t)id1 * S.tri_rec_bytes, C_TRI)); ++*n_loads;
                hit = S.tri(id1, r.o, r.d, r.tmin, r.tmax, best) || hit;
            }
        }
        if (hit || best != FLT_MAX) return;
        if (sp == 0) return;
        const E& e = st[--sp];
        r.tmin = e.tmin; r.tmax = e.tmax;
        if (e.w1 & 4u) {
            if (e.has_rec) memcpy(rec, e.rec, 16);
            else load_rec(e.w1 >> 3, rec, C_POP);
            lvl = 0;
        } else { rec[0] = e.w0; rec[1] = e.w1; lvl = 1; }
    }
}

// ---- pair-addressed nodes (layouts 0 and 1), as trace_kernel<!TRIPLES> walks them ----
static void walk_pairs(const Scene& S, const gigl::DeviceTree& D, const Ray& r0, std::vector<uint64_t>& seq, uint64_t* n_loads) {
    Ray r = r0;
    const std::vector<KdNode>& N = D.slots;
    const uint32_t* ids = D.ids_inline ? reinterpret_cast<const uint32_t*>(N.data()) : D.ids.data();
    const uint64_t ids_region = D.ids_inline ? 0 : (2ull << 40);
    struct E { float tmin, tmax; uint32_t w0, w1; } st[64];
    int sp = 0;
    uint32_t w0 = N[0].w0, w1 = N[0].w1;
    seq.push_back(0);
    float best = FLT_MAX;
    for (;;) {
        if (best < r.tmin) return;
        while ((int32_t)w1 >= 0) {
            const uint32_t pair = w1 >> 2;
            seq.push_back(tag((uint64_t)pair * 8u, C_DESC)); ++*n_loads;
            const Step s = decide(r, w1 & 3u, asf(w0));
            const KdNode& a = N[pair], &b = N[pair + 1];
            if (s.both) {
                E& e = st[sp++];
                e.tmin = s.ts - eps_of(s.ts); e.tmax = r.tmax;
                e.w0 = s.bottom_first ? b.w0 : a.w0; e.w1 = s.bottom_first ? b.w1 : a.w1;
                r.tmax = s.ts + eps_of(s.ts);
            }
            if (s.take_top) { w0 = b.w0; w1 = b.w1; } else { w0 = a.w0; w1 = a.w1; }
        }
        const uint32_t kind = (w1 >> 29) & 3u;
        const uint32_t n = kind == 3u ? (w1 & 0x1FFFFFFFu) : kind;
        bool hit = false;
        for (uint32_t i = 0; i < n; i += 2) {
            uint32_t id0, id1;
            if (n >= 3u) {
                seq.push_back(tag(ids_region | ((uint64_t)(w0 + i) * 4u), C_IDS)); ++*n_loads;
                id0 = ids[w0 + i]; id1 = i + 1 < n ? ids[w0 + i + 1] : id0;
            } else { id0 = w0; id1 = w1 & 0x1FFFFFFFu; }
            seq.push_back(tag(kTriRegion | (uint64_t)id0 * S.tri_rec_bytes, C_TRI)); ++*n_loads;
            hit = S.tri(id0, r.o, r.d, r.tmin, r.tmax, best) || hit;
            if (i + 1 < n) {
                seq.push_back(tag(kTriRegion | (uint64_t)id1 * S.tri_rec_bytes, C_TRI)); ++*n_loads;
                hit = S.tri(id1, r.o, r.d, r.tmin, r.tmax, best) || hit;
            }
        }
        if (hit || best != FLT_MAX) return;
        if (sp == 0) return;
        const E& e = st[--sp];
        r.tmin = e.tmin; r.tmax = e.tmax; w0 = e.w0; w1 = e.w1;
    }
}

// ---- the same walk over the HOST tree, emitting abstract accesses (record of host node k, leaf record of
// leaf k, id array of leaf k) that a table maps to addresses: prices any assignment of records to lines,
// encodable or not.  Record roots are the nodes at even depth. ----
struct Abstract {
    std::vector<uint64_t> rec, leafrec, ids;   // byte address per host node (~0 = none)
    std::vector<uint64_t> block;               // optional: address of the leaf's private triangle block (40-byte copies, 3 per line)
    uint32_t block_min = 2;                    // leaves with at least this many prims use their block
};
static void walk_host(const Scene& S, const gigl::KdTreeHost& T, const Abstract& A, const Ray& r0, bool by_value,
                      std::vector<uint64_t>& seq) {
    const std::vector<KdNode>& H = T.nodes;
    Ray r = r0;
    struct E { float tmin, tmax; uint32_t node; int depth; bool loaded; } st[64];
    int sp = 0;
    uint32_t k = 0;
    int depth = 0;
    float best = FLT_MAX;
    seq.push_back(tag(A.rec[0], C_ROOT));
    for (;;) {
        if (best < r.tmin) return;
        while (!(H[k].w1 & 1u)) {
            const Step s = decide(r, (H[k].w1 >> 1) & 3u, asf(H[k].w0));
            const uint32_t below = k + 1u, above = H[k].w1 >> 3;
            const uint32_t near_k = s.take_top ? above : below;
            if (s.both) {
                const uint32_t far_k = s.bottom_first ? above : below;
                E& e = st[sp++];
                e.tmin = s.ts - eps_of(s.ts); e.tmax = r.tmax; e.node = far_k; e.depth = depth + 1; e.loaded = false;
                if (by_value && (depth & 1)) { seq.push_back(tag(A.rec[far_k], C_FARVAL)); e.loaded = true; }
                r.tmax = s.ts + eps_of(s.ts);
            }
            k = near_k;
            ++depth;
            if (!(depth & 1)) seq.push_back(tag(A.rec[k], C_DESC));
        }
        const uint32_t n = H[k].w1 >> 1;
        if ((depth & 1) && n >= 2) seq.push_back(tag(A.leafrec[k], C_LEAFREC));
        bool hit = false;
        const bool blk = !A.block.empty() && n >= A.block_min && A.block[k] != ~0ull;
        auto tri_addr = [&](uint32_t i, uint32_t id) -> uint64_t {
            if (blk) return (2ull << 40) | (A.block[k] + (uint64_t)(i / 3u) * 128u + (uint64_t)(i % 3u) * 40u);
            return kTriRegion | (uint64_t)id * S.tri_rec_bytes;
        };
        for (uint32_t i = 0; i < n; i += 2) {
            if (n > 3u && !blk) seq.push_back(tag(A.ids[k] + (uint64_t)(i >> 1) * 8u, C_IDS));
            const uint32_t id0 = n == 1 ? H[k].w0 : T.leaf_prim_ids[H[k].w0 + i];
            seq.push_back(tag(tri_addr(i, id0), C_TRI));
            hit = S.tri(id0, r.o, r.d, r.tmin, r.tmax, best) || hit;
            if (i + 1 < n) {
                const uint32_t id1 = T.leaf_prim_ids[H[k].w0 + i + 1];
                seq.push_back(tag(tri_addr(i + 1, id1), C_TRI));
                hit = S.tri(id1, r.o, r.d, r.tmin, r.tmax, best) || hit;
            }
        }
        if (hit || best != FLT_MAX) return;
        if (sp == 0) return;
        const E& e = st[--sp];
        r.tmin = e.tmin; r.tmax = e.tmax; k = e.node; depth = e.depth;
        if (!(depth & 1) && !e.loaded) seq.push_back(tag(A.rec[k], C_POP));
    }
}

// Free-form greedy clusters over the record tree (a record = a node at even depth with its two children;
// its child records are its up to four interior-or-leaf grandchildren): a cluster grows from its root by
// largest cell surface area while the 128-byte line has room (8 units of 16 bytes); a record costs one unit
// plus the units of the leaf records / id arrays of its leaf children, which go with it.  No constraint on
// where sibling records go.  min_fill: a cluster root starts a new line unless its whole subtree fits in
// what is left of the current one.
static Abstract cluster_free(const gigl::KdTreeHost& T, size_t* bytes_out, uint32_t rec_dw = 4, bool pairs_adjacent = false,
                             uint32_t line_dw = 32, bool own_line = false) {
    const std::vector<KdNode>& H = T.nodes;
    const size_t n = H.size();
    Abstract A;
    A.rec.assign(n, ~0ull); A.leafrec.assign(n, ~0ull); A.ids.assign(n, ~0ull);
    auto is_leaf = [&](uint32_t k) { return (H[k].w1 & 1u) != 0; };
    auto n_of = [&](uint32_t k) { return H[k].w1 >> 1; };
    // everything in dwords; a line is 32 dwords; leaf records are 4 dwords (count + up to 3 ids, or the
    // array's address), id arrays n dwords rounded up to 2
    auto ids_units = [&](uint32_t c) { return (c + 1u) & ~1u; };
    auto cost1 = [&](uint32_t k) -> uint32_t {
        if (is_leaf(k)) return 4u + (n_of(k) > 3u ? ids_units(n_of(k)) : 0u);
        uint32_t c = rec_dw;
        const uint32_t kids[2] = {k + 1u, H[k].w1 >> 3};
        for (uint32_t ch : kids)
            if (is_leaf(ch) && n_of(ch) >= 2u) c += 4u + (n_of(ch) > 3u ? ids_units(n_of(ch)) : 0u);
        return c;
    };
    // pairs_adjacent: the two child records of a lvl-1 node are placed together (one pointer per pair);
    // a candidate is then the pair, named by its first member (the bottom child, k), partner = sibling
    std::vector<uint32_t> partner(n, ~0u);
    auto cost = [&](uint32_t k) -> uint32_t { return cost1(k) + (partner[k] != ~0u ? cost1(partner[k]) : 0u); };
    auto child_recs = [&](uint32_t k, uint32_t g[4]) -> uint32_t {
        uint32_t m = 0;
        if (is_leaf(k)) return 0;
        const uint32_t kids[2] = {k + 1u, H[k].w1 >> 3};
        for (uint32_t ch : kids)
            if (!is_leaf(ch)) { g[m++] = ch + 1u; g[m++] = H[ch].w1 >> 3; }
        return m;
    };
    // subtree units (saturating) for record roots
    std::vector<uint32_t> sub(n, 0u);
    std::vector<uint8_t> is_rec(n, 0);
    is_rec[0] = 1;
    for (uint32_t k = 0; k < n; ++k) {
        if (!is_rec[k]) continue;
        uint32_t g[4];
        const uint32_t m = child_recs(k, g);
        for (uint32_t i = 0; i < m; ++i) is_rec[g[i]] = 1;
    }
    for (size_t k = n; k-- > 0;) {
        if (!is_rec[k]) continue;
        uint32_t g[4];
        const uint32_t m = child_recs((uint32_t)k, g);
        uint64_t s = cost1((uint32_t)k);
        for (uint32_t i = 0; i < m; ++i) s += sub[g[i]];
        sub[k] = s > 1000000 ? 1000000u : (uint32_t)s;
    }
    struct Cand { float sa; uint32_t k; float b[6]; };
    auto area = [](const float* b) { const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2]; return dx * dy + dy * dz + dz * dx; };
    auto clip = [&](float* b, uint32_t parent, int j) {
        const unsigned axis = (H[parent].w1 >> 1) & 3u;
        const float split = asf(H[parent].w0);
        if (j == 0) b[3 + axis] = split; else b[axis] = split;
    };
    const uint32_t kLineU = line_dw;   // usable dwords per 128-byte line (a header takes the rest)
    if (pairs_adjacent)
        for (uint32_t k = 0; k < n; ++k)
            if (!is_leaf(k)) { /* children of k: k+1 and above; if they are record roots (odd-depth parent) pair them */ }
    uint64_t line = 0;       // current line index
    uint32_t used = kLineU;  // force a new line at the start
    uint64_t n_lines = 0;
    auto place1 = [&](uint32_t k) {
        const uint64_t base = line * 128u + (uint64_t)used * 4u;
        A.rec[k] = base;
        uint32_t off = is_leaf(k) ? 4u : rec_dw;
        if (is_leaf(k)) { if (n_of(k) > 3u) { A.ids[k] = base + 16u; off += ids_units(n_of(k)); } }
        else {
            const uint32_t kids[2] = {k + 1u, H[k].w1 >> 3};
            for (uint32_t ch : kids)
                if (is_leaf(ch) && n_of(ch) >= 2u) {
                    A.leafrec[ch] = base + (uint64_t)off * 4u; off += 4u;
                    if (n_of(ch) > 3u) { A.ids[ch] = base + (uint64_t)off * 4u; off += ids_units(n_of(ch)); }
                }
        }
        used += off;
    };
    auto place = [&](uint32_t k, uint32_t c) {   // c dwords at the current line
        const uint32_t before = used;
        place1(k);
        if (partner[k] != ~0u) place1(partner[k]);
        if (used > before + c) used = before + c;   // oversize arrays were capped
        if (used > kLineU) used = kLineU;
    };
    std::vector<Cand> roots;
    {
        Cand c; c.k = 0;
        for (int a = 0; a < 3; ++a) { c.b[a] = T.box_min[a]; c.b[3 + a] = T.box_max[a]; }
        c.sa = area(c.b);
        roots.push_back(c);
    }
    std::vector<Cand> heap;
    auto cmp = [](const Cand& a, const Cand& b) { return a.sa < b.sa; };
    while (!roots.empty()) {
        const Cand r = roots.back();
        roots.pop_back();
        const uint32_t free_u = kLineU - used;
        const uint32_t need = sub[r.k] + (partner[r.k] != ~0u ? sub[partner[r.k]] : 0u);
        if (own_line || !(need <= free_u)) { line = n_lines++; used = 0; }   // own_line: every cluster root starts a line
        heap.clear();
        heap.push_back(r);
        std::vector<Cand> deferred;
        while (!heap.empty()) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            const Cand c = heap.back();
            heap.pop_back();
            uint32_t cu = cost(c.k);
            if (cu > kLineU) cu = kLineU;   // oversize id arrays spill (not modelled further)
            if (cu > kLineU - used) { deferred.push_back(c); continue; }
            place(c.k, cu);
            for (int member = 0; member < 2; ++member) {
            const uint32_t ck = member == 0 ? c.k : partner[c.k];
            if (ck == ~0u || is_leaf(ck)) continue;
            Cand cc = c; cc.k = ck;
            const Cand& c = cc;
            const uint32_t kids[2] = {c.k + 1u, H[c.k].w1 >> 3};
            for (int j = 0; j < 2; ++j) {
                if (is_leaf(kids[j])) continue;
                const uint32_t gk[2] = {kids[j] + 1u, H[kids[j]].w1 >> 3};
                for (int jj = 0; jj < 2; ++jj) {
                    Cand nx; nx.k = gk[jj];
                    memcpy(nx.b, c.b, sizeof nx.b);
                    clip(nx.b, c.k, j);
                    if (pairs_adjacent) { if (jj == 1) continue; partner[gk[0]] = gk[1]; }   // the pair's cell = the child's cell
                    else clip(nx.b, kids[j], jj);
                    nx.sa = area(nx.b);
                    heap.push_back(nx);
                    std::push_heap(heap.begin(), heap.end(), cmp);
                }
            }
            }
        }
        // small subtrees first (they may share the next line), large ones after, by area
        std::stable_sort(deferred.begin(), deferred.end(), [&](const Cand& a, const Cand& b) {
            const bool sa_ = sub[a.k] <= kLineU, sb_ = sub[b.k] <= kLineU;
            if (sa_ != sb_) return sa_;
            return a.sa > b.sa;
        });
        for (size_t i = deferred.size(); i-- > 0;) roots.push_back(deferred[i]);
    }
    (void)cost1;
    *bytes_out = (size_t)n_lines * 128u;
    return A;
}

struct Priced { double loads, switches, miss[5], tri_miss, distinct, cat_loads[C_N], cat_miss[C_N]; };

static Priced price(const std::vector<std::vector<uint64_t>>& seqs, double hot_frac) {
    std::unordered_map<uint64_t, uint32_t> freq;
    std::vector<uint64_t> tmp;
    for (const auto& s : seqs) {
        tmp.clear();
        for (uint64_t a : s) tmp.push_back((a & 0xFFFFFFFFFFFull) >> 7);
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        for (uint64_t l : tmp) ++freq[l];
    }
    const uint32_t thr = (uint32_t)std::max(1.0, hot_frac * (double)seqs.size());
    Priced P = {};
    for (const auto& s : seqs) {
        uint64_t lru[5][5];
        int nl[5] = {0, 0, 0, 0, 0};
        uint64_t prev = ~0ull;
        tmp.clear();
        for (uint64_t a : s) {
            const uint64_t l = (a & 0xFFFFFFFFFFFull) >> 7;
            const int cat = (int)(a >> 44) & 7;
            P.cat_loads[cat] += 1;
            P.loads += 1;
            if (freq[l] <= thr) tmp.push_back(l);
            if (l != prev) P.switches += 1;
            prev = l;
            const bool hot = freq[l] > thr;
            for (int K = 1; K <= 4; ++K) {
                int found = -1;
                for (int i = 0; i < nl[K]; ++i) if (lru[K][i] == l) found = i;
                if (found < 0) {
                    if (!hot) { P.miss[K] += 1; if (K == 2) { P.cat_miss[cat] += 1; if (a & kTriRegion) P.tri_miss += 1; } }
                    if (nl[K] < K) ++nl[K];
                    for (int i = nl[K] - 1; i > 0; --i) lru[K][i] = lru[K][i - 1];
                } else {
                    for (int i = found; i > 0; --i) lru[K][i] = lru[K][i - 1];
                }
                lru[K][0] = l;
            }
        }
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        P.distinct += (double)tmp.size();
    }
    const double n = (double)seqs.size();
    P.loads /= n; P.switches /= n; P.tri_miss /= n; P.distinct /= n;
    for (int K = 1; K <= 4; ++K) P.miss[K] /= n;
    for (int c = 0; c < C_N; ++c) { P.cat_loads[c] /= n; P.cat_miss[c] /= n; }
    return P;
}

static void show(const char* name, const Priced& P, size_t bytes) {
    printf("%-44s loads %6.2f  misses/ray K=1 %6.2f K=2 %6.2f (tri %5.2f) K=3 %6.2f K=4 %6.2f | cold distinct %6.2f | %7.1f MB\n", name,
           P.loads, P.miss[1], P.miss[2], P.tri_miss, P.miss[3], P.miss[4], P.distinct, bytes / 1e6);
    printf("      by kind (loads / K=2 misses per ray):");
    for (int c = 0; c < C_N; ++c) if (P.cat_loads[c] > 0) printf("  %s %.2f/%.2f", kCatName[c], P.cat_loads[c], P.cat_miss[c]);
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp/layout_sim";
    const double hot_frac = argc > 2 ? atof(argv[2]) : 5e-4;
    auto hdr = slurp(dir + "/header.bin");
    const uint32_t* h = (const uint32_t*)hdr.data();
    const uint32_t nv = h[0], nt = h[1], nr = h[2];
    auto vb = slurp(dir + "/verts.bin"), tb = slurp(dir + "/tris.bin"), ob = slurp(dir + "/ro.bin"), db = slurp(dir + "/rd.bin");
    Scene S;
    S.V = (const float*)vb.data();
    S.I = (const uint32_t*)tb.data();
    gigl::KdTreeHost T;
    gigl::KdBuildParams bp;
    const char* err = nullptr;
    if (!gigl::kd_build(S.V, nv, S.I, nt, bp, T, &err)) { fprintf(stderr, "build: %s\n", err ? err : "?"); return 1; }
    for (int a = 0; a < 3; ++a) { S.box_min[a] = T.box_min[a]; S.box_max[a] = T.box_max[a]; }
    printf("tree: %zu nodes, depth %d; %u rays; hot = lines touched by more than %.2g of the rays\n", T.nodes.size(),
           T.actual_depth, nr, hot_frac);
    std::vector<Ray> rays;
    for (uint32_t i = 0; i < nr; ++i) {
        Ray r;
        if (setup(S, (const float*)ob.data() + 4 * (size_t)i, (const float*)db.data() + 4 * (size_t)i, r)) rays.push_back(r);
    }
    std::vector<std::vector<uint64_t>> seqs(rays.size());
    auto reset = [&]() { for (auto& s : seqs) s.clear(); };
    const bool quick = getenv("SEQ_QUICK") != nullptr;
    if (!quick) {
        gigl::DeviceTree D;
        gigl::layout_clustered(T, D);
        uint64_t nl = 0;
        reset();
        for (size_t i = 0; i < rays.size(); ++i) walk_pairs(S, D, rays[i], seqs[i], &nl);
        show("layout 1 (clustered pairs), tris 64 B", price(seqs, hot_frac), D.slots.size() * 8);
    }
    {
        gigl::DeviceTree D;
        if (!gigl::layout_triples(T, D)) { printf("triples: does not fit\n"); return 0; }
        const uint32_t* W = reinterpret_cast<const uint32_t*>(D.slots.data());
        for (int bv = 0; bv < (quick ? 1 : 2); ++bv) {
            uint64_t nl = 0;
            reset();
            for (size_t i = 0; i < rays.size(); ++i) walk_triples(S, W, rays[i], bv != 0, seqs[i], &nl);
            show(bv ? "layout 2 (triples), far record by value" : "layout 2 (triples), tris 64 B", price(seqs, hot_frac), D.slots.size() * 8);
        }
        if (quick) return 0;
        {   // the real layout through the host walk (must reproduce the numbers of the exact walk above)
            gigl::DeviceTree D2;
            std::vector<uint32_t> slot_of, list_dw;
            D2.slot_of_host = &slot_of; D2.list_dw_of_host = &list_dw;
            gigl::layout_triples(T, D2);
            const uint32_t* W2 = reinterpret_cast<const uint32_t*>(D2.slots.data());
            Abstract A;
            const size_t nn = T.nodes.size();
            A.rec.assign(nn, ~0ull); A.leafrec.assign(nn, ~0ull); A.ids.assign(nn, ~0ull);
            for (size_t k = 0; k < nn; ++k) {
                A.rec[k] = (uint64_t)slot_of[k] * 8u;
                const bool leaf = T.nodes[k].w1 & 1u;
                const uint32_t cnt = T.nodes[k].w1 >> 1;
                if (leaf && list_dw[k] != ~0u) {
                    // list_dw: dword address of the leaf's extra record (n <= 3 at a child position) or of its id array
                    if (cnt > 3u) { A.ids[k] = (uint64_t)list_dw[k] * 4u; A.leafrec[k] = ~0ull; }
                    else A.leafrec[k] = (uint64_t)list_dw[k] * 4u;
                }
            }
            // leaf records of child-position leaves with n > 3: the record that points at the id array
            // (address not exported by the layout: take the array's line, where alloc_aux puts both when it can)
            for (size_t k = 0; k < nn; ++k)
                if ((T.nodes[k].w1 & 1u) && (T.nodes[k].w1 >> 1) > 3u && A.leafrec[k] == ~0ull) A.leafrec[k] = A.ids[k];
            (void)W2;
            reset();
            for (size_t i = 0; i < rays.size(); ++i) walk_host(S, T, A, rays[i], false, seqs[i]);
            show("layout 2 through the host walk (check)", price(seqs, hot_frac), D2.slots.size() * 8);
            for (uint32_t bmin : {2u, 3u, 4u}) {
                A.block.assign(nn, ~0ull);
                A.block_min = bmin;
                uint64_t next = 0;
                for (size_t k = 0; k < nn; ++k)
                    if ((T.nodes[k].w1 & 1u) && (T.nodes[k].w1 >> 1) >= bmin) { A.block[k] = next; next += (uint64_t)(((T.nodes[k].w1 >> 1) + 2u) / 3u) * 128u; }
                reset();
                for (size_t i = 0; i < rays.size(); ++i) walk_host(S, T, A, rays[i], false, seqs[i]);
                char nm[96];
                snprintf(nm, sizeof nm, "layout 2 + leaf triangle blocks (n >= %u)", bmin);
                show(nm, price(seqs, hot_frac), D2.slots.size() * 8 + next);
            }
        }
        struct V { const char* name; uint32_t dw; bool adj; bool bv; uint32_t line_dw; bool own; };
        const V vars[] = {{"free-form, 16 B records (bound)", 4, false, false, 32, false},
                          {"free-form 16 B, 7 units + header, root per line", 4, false, false, 28, true},
                          {"free-form 16 B, 7 units + header, shared lines", 4, false, false, 28, false},
                          {"free-form, 20 B records", 5, false, false, 32, false},
                          {"pairs adjacent, 16 B (bound)", 4, true, false, 32, false}};
        for (const V& v : vars) {
            size_t bytes = 0;
            const Abstract A = cluster_free(T, &bytes, v.dw, v.adj, v.line_dw, v.own);
            reset();
            for (size_t i = 0; i < rays.size(); ++i) walk_host(S, T, A, rays[i], v.bv, seqs[i]);
            show(v.name, price(seqs, hot_frac), bytes);
        }
        S.tri_rec_bytes = 32;
        uint64_t nl = 0;
        reset();
        for (size_t i = 0; i < rays.size(); ++i) walk_triples(S, W, rays[i], false, seqs[i], &nl);
        show("layout 2 (triples), tris 32 B (pair records)", price(seqs, hot_frac), D.slots.size() * 8);
        S.tri_rec_bytes = 64;
    }
    return 0;
}
