// kd_layout.cpp — device addressing of the host k-d tree (see kd_layout.h).
#include "kd_layout.h"

#include <algorithm>
#include <cstring>
#include <queue>

namespace gigl {

namespace {

inline bool is_leaf(const KdNode& h) { return (h.w1 & 1u) != 0u; }

// Re-encodes one host leaf for the device: leaves get a kind field and 2-prim leaves carry their
// ids inline.  Kind-3 leaves keep w0 = offset into leaf_prim_ids (the caller may redirect it).
inline KdNode device_leaf(const KdTreeHost& T, const KdNode& h) {
    KdNode nd = h;
    const uint32_t np = h.w1 >> 1;
    if (np == 0) { nd.w0 = 0; nd.w1 = kDevLeafBit; }
    else if (np == 1) nd.w1 = kDevLeafBit | (1u << 29);
    else if (np == 2) {
        nd.w0 = T.leaf_prim_ids[h.w0];
        nd.w1 = kDevLeafBit | (2u << 29) | T.leaf_prim_ids[h.w0 + 1];
    } else nd.w1 = kDevLeafBit | (3u << 29) | np;
    return nd;
}

}  // namespace

// Device node order of round 1.  The reference numbers nodes depth-first (bottom child = parent + 1,
// KdTree.hpp:254-262), which makes every step to a top child a jump to a far cache line.  Here the
// two children of a node sit side by side (one 16-byte pair; the parent stores the pair index) and
// pairs are laid out in treelets: a breadth-first walk of `levels` levels of pairs below a treelet
// root (1 + 2 + 4 = 7 pairs = 112 B for 3 levels) is stored contiguously and never straddles a
// 128-byte line, so up to three consecutive steps of a ray read one line.  Topology, leaf contents
// and traversal order are untouched; only addresses change.  Slot 0 is the root, slot 1 padding.
void layout_levels(const KdTreeHost& T, int levels, DeviceTree& dt) {
    std::vector<KdNode>& out = dt.slots;
    dt.ids_inline = false;
    dt.ids.clear();
    out.clear();
    auto dev_leaf = [&](const KdNode& h) {
        KdNode nd = device_leaf(T, h);
        const uint32_t n = h.w1 >> 1;
        if (n >= 3u) {   // copy the list to an even offset, padded to an even length
            nd.w0 = (uint32_t)dt.ids.size();
            for (uint32_t i = 0; i < n; ++i) dt.ids.push_back(T.leaf_prim_ids[h.w0 + i]);
            if (n & 1u) dt.ids.push_back(0u);
        }
        return nd;
    };
    const std::vector<KdNode>& H = T.nodes;
    if (dt.slot_of_host) dt.slot_of_host->assign(H.size(), 0u);
    if (H.empty()) {  // empty scene: a single empty leaf (KdTree.hpp:9-12)
        out.push_back(KdNode{0u, kDevLeafBit});
        out.push_back(KdNode{0u, kDevLeafBit});
        return;
    }
    out.reserve(H.size() + H.size() / 4 + 16);
    out.resize(2, KdNode{0u, kDevLeafBit});
    const unsigned max_pairs = (1u << levels) - 1u;           // pairs in a full treelet
    const unsigned line_pairs = 8;                            // 128 B / 16 B
    struct Item { uint32_t host; uint32_t slot; };            // interior host node, its device slot
    std::vector<Item> roots, cur, nxt;
    if (is_leaf(H[0])) { out[0] = dev_leaf(H[0]); return; }
    out[0] = KdNode{H[0].w0, (H[0].w1 >> 1) & 3u};
    roots.push_back(Item{0u, 0u});
    while (!roots.empty()) {
        const Item r = roots.back();
        roots.pop_back();
        // first pass: how many pairs will this treelet hold?
        unsigned n_pairs = 0;
        cur.assign(1, r);
        for (int lv = 0; lv < levels && !cur.empty(); ++lv) {
            nxt.clear();
            for (const Item& it : cur) {
                ++n_pairs;
                const uint32_t kids[2] = {it.host + 1u, H[it.host].w1 >> 3};
                for (uint32_t k : kids)
                    if (!is_leaf(H[k])) nxt.push_back(Item{k, 0u});
            }
            cur.swap(nxt);
        }
        size_t pos = out.size() / 2;                           // in pairs
        if (max_pairs <= line_pairs && (pos % line_pairs) + n_pairs > line_pairs)
            pos = (pos + line_pairs - 1) / line_pairs * line_pairs;
        out.resize(2 * pos, KdNode{0u, kDevLeafBit});
        // second pass: emit breadth-first
        cur.assign(1, r);
        std::vector<Item> frontier;
        for (int lv = 0; lv < levels && !cur.empty(); ++lv) {
            nxt.clear();
            for (const Item& it : cur) {
                const uint32_t pair = (uint32_t)(out.size() / 2);
                out[it.slot].w1 = (out[it.slot].w1 & 3u) | ((uint32_t)(2 * pair) << 2);
                const uint32_t kids[2] = {it.host + 1u, H[it.host].w1 >> 3};
                for (uint32_t k : kids) {
                    const uint32_t slot = (uint32_t)out.size();
                    if (dt.slot_of_host) (*dt.slot_of_host)[k] = slot;
                    if (is_leaf(H[k])) out.push_back(dev_leaf(H[k]));
                    else {
                        out.push_back(KdNode{H[k].w0, (H[k].w1 >> 1) & 3u});
                        (lv + 1 < levels ? nxt : frontier).push_back(Item{k, slot});
                    }
                }
            }
            cur.swap(nxt);
        }
        // deeper treelets of this subtree follow it closely (depth-first over treelets)
        for (size_t i = frontier.size(); i-- > 0;) roots.push_back(frontier[i]);
    }
}

// Clustered layout.  The buffer is a sequence of 128-byte lines of 32 dwords.  Within a line child
// pairs (4 dwords, 16-byte aligned) grow from the front and the id lists of the line's kind-3
// leaves (n dwords) grow from the back; a list that cannot share a line with its leaf's pair goes
// to an overflow area behind the last line.  A cluster starts at an interior node whose child pair
// is not placed yet and adds pairs in the order of their parents' cell surface area.
void layout_clustered(const KdTreeHost& T, DeviceTree& dt) {
    constexpr uint32_t kLine = 32;                             // dwords per 128-byte line
    dt.ids_inline = true;
    dt.slots.clear();
    const std::vector<KdNode>& H = T.nodes;
    if (dt.slot_of_host) dt.slot_of_host->assign(H.size(), 0u);
    std::vector<uint32_t> W;                                   // the buffer, as dwords
    auto put_node = [&](uint32_t at_dw, KdNode nd) { W[at_dw] = nd.w0; W[at_dw + 1] = nd.w1; };
    const KdNode pad{0u, kDevLeafBit};
    if (H.empty() || is_leaf(H[0])) {
        W.assign(4, 0u);
        put_node(0, H.empty() ? pad : device_leaf(T, H[0]));
        put_node(2, pad);
        if (!H.empty() && (H[0].w1 >> 1) >= 3u) {              // a root leaf with an id list
            const uint32_t n = H[0].w1 >> 1;
            W[0] = (uint32_t)W.size();
            for (uint32_t i = 0; i < n; ++i) W.push_back(T.leaf_prim_ids[H[0].w0 + i]);
            if (W.size() & 1u) W.push_back(0u);
        }
        dt.slots.resize(W.size() / 2);
        std::memcpy(dt.slots.data(), W.data(), W.size() * 4);
        return;
    }
    const uint32_t n_nodes = (uint32_t)H.size();
    // dwords a pair adds to its line: the pair itself and the id lists of its kind-3 leaves (lists
    // too long to share a line with the pair are left to the overflow area)
    auto list_dw = [&](uint32_t k) -> uint32_t {
        if (!is_leaf(H[k])) return 0u;
        const uint32_t n = H[k].w1 >> 1;
        return n >= 3u ? ((n + 1u) & ~1u) : 0u;               // padded to an even length (8-byte loads)
    };
    auto pair_cost = [&](uint32_t host, uint32_t& l0, uint32_t& l1) -> uint32_t {
        l0 = list_dw(host + 1u);
        l1 = list_dw(H[host].w1 >> 3);
        uint32_t c = 4u;
        if (c + l0 <= kLine) c += l0; else l0 = ~0u;           // ~0: overflow
        if (l1 && c + l1 <= kLine) c += l1; else if (l1) l1 = ~0u;
        return c;
    };
    // dwords of the whole subtree below an interior node (saturating), children after parents in
    // the reference's numbering, so a reverse sweep sees children first
    std::vector<uint32_t> sub(n_nodes, 0u);
    for (uint32_t i = n_nodes; i-- > 0;) {
        if (is_leaf(H[i])) continue;
        uint32_t l0, l1;
        uint64_t s = pair_cost(i, l0, l1);
        s += sub[i + 1u];
        s += sub[H[i].w1 >> 3];
        sub[i] = s > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)s;
    }
    struct Cand { float sa; uint32_t host; uint32_t at_dw; float b[6]; };   // interior node placed at at_dw
    struct BySa { bool operator()(const Cand& a, const Cand& b) const { return a.sa < b.sa; } };
    auto area = [](const float* b) {
        const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
        return dx * dy + dy * dz + dz * dx;
    };
    W.reserve((size_t)n_nodes * 2 + T.leaf_prim_ids.size() + n_nodes / 2);
    W.assign(kLine, 0u);
    for (uint32_t i = 0; i < kLine; i += 2) put_node(i, pad);
    put_node(0, KdNode{H[0].w0, (H[0].w1 >> 1) & 3u});
    uint32_t line = 0, front = 4, back = kLine;                // current line: dword offsets within it
    auto new_line = [&]() {
        line = (uint32_t)W.size();
        W.resize(W.size() + kLine, 0u);
        for (uint32_t i = 0; i < kLine; i += 2) put_node(line + i, pad);
        front = 0; back = kLine;
    };
    struct Overflow { uint32_t at_dw; uint32_t host; };
    std::vector<Overflow> overflow;
    std::vector<Cand> roots, deferred;
    {
        Cand c;
        c.host = 0; c.at_dw = 0;
        for (int a = 0; a < 3; ++a) { c.b[a] = T.box_min[a]; c.b[3 + a] = T.box_max[a]; }
        c.sa = area(c.b);
        roots.push_back(c);
    }
    std::priority_queue<Cand, std::vector<Cand>, BySa> heap;
    while (!roots.empty()) {
        const Cand r = roots.back();
        roots.pop_back();
        uint32_t l0, l1;
        const uint32_t first_cost = pair_cost(r.host, l0, l1);
        const uint32_t free_dw = back - front;
        // a subtree that fits in one line is kept whole; a larger one starts a line of its own
        // unless most of a line is still free
        if (sub[r.host] <= kLine ? sub[r.host] > free_dw : (free_dw < kLine / 2 || first_cost > free_dw)) new_line();
        heap.push(r);
        deferred.clear();
        while (!heap.empty()) {
            const Cand c = heap.top();
            heap.pop();
            const uint32_t cost = pair_cost(c.host, l0, l1);
            if (cost > back - front) { deferred.push_back(c); continue; }
            const uint32_t at = line + front;
            front += 4;
            W[c.at_dw + 1] = (W[c.at_dw + 1] & 3u) | ((at / 2u) << 2);
            const uint32_t kids[2] = {c.host + 1u, H[c.host].w1 >> 3};
            const uint32_t lists[2] = {l0, l1};
            const unsigned axis = (H[c.host].w1 >> 1) & 3u;
            float split;
            std::memcpy(&split, &H[c.host].w0, 4);
            for (int j = 0; j < 2; ++j) {
                const uint32_t k = kids[j], kd = at + 2u * (uint32_t)j;
                if (dt.slot_of_host) (*dt.slot_of_host)[k] = kd / 2u;
                if (is_leaf(H[k])) {
                    KdNode nd = device_leaf(T, H[k]);
                    if (lists[j] == ~0u) overflow.push_back(Overflow{kd, k});
                    else if (lists[j]) {
                        back -= lists[j];
                        nd.w0 = line + back;
                        const uint32_t n = H[k].w1 >> 1;
                        for (uint32_t i = 0; i < n; ++i) W[line + back + i] = T.leaf_prim_ids[H[k].w0 + i];
                        if (n & 1u) W[line + back + n] = 0u;
                    }
                    put_node(kd, nd);
                } else {
                    put_node(kd, KdNode{H[k].w0, (H[k].w1 >> 1) & 3u});
                    Cand n;
                    n.host = k; n.at_dw = kd;
                    std::memcpy(n.b, c.b, sizeof(n.b));
                    if (j == 0) n.b[3 + axis] = split; else n.b[axis] = split;   // bottom child: below the plane
                    n.sa = area(n.b);
                    heap.push(n);
                }
            }
        }
        // clusters still to place: small subtrees first (they fill what is left of this line),
        // larger ones after them; the stack pops in reverse push order
        std::stable_sort(deferred.begin(), deferred.end(), [&](const Cand& a, const Cand& b) {
            const bool sa_ = sub[a.host] <= kLine, sb_ = sub[b.host] <= kLine;
            if (sa_ != sb_) return sa_;
            return a.sa > b.sa;
        });
        for (size_t i = deferred.size(); i-- > 0;) roots.push_back(deferred[i]);
    }
    for (const Overflow& o : overflow) {
        const uint32_t n = H[o.host].w1 >> 1;
        if (W.size() & 1u) W.push_back(0u);
        W[o.at_dw] = (uint32_t)W.size();
        for (uint32_t i = 0; i < n; ++i) W.push_back(T.leaf_prim_ids[H[o.host].w0 + i]);
    }
    W.push_back(0u);   // the last list may be read one id past its end
    if (W.size() & 1u) W.push_back(0u);
    dt.slots.resize(W.size() / 2);
    std::memcpy(dt.slots.data(), W.data(), W.size() * 4);
}

bool layout_verify(const KdTreeHost& T, const DeviceTree& D, const char** what) {
    auto bad = [&](const char* m) { if (what) *what = m; return false; };
    const std::vector<KdNode>& H = T.nodes;
    const std::vector<KdNode>& S = D.slots;
    if (S.size() < 2) return bad("device tree has no root pair");
    if (H.empty()) return S[0].w1 == kDevLeafBit ? true : bad("empty tree must be an empty leaf");
    const uint32_t* ids = D.ids_inline ? reinterpret_cast<const uint32_t*>(S.data()) : D.ids.data();
    const size_t n_ids = D.ids_inline ? S.size() * 2 : D.ids.size();
    struct It { uint32_t host, slot; };
    std::vector<It> st(1, It{0u, 0u});
    size_t seen = 0;
    while (!st.empty()) {
        const It it = st.back();
        st.pop_back();
        ++seen;
        if (it.slot >= S.size()) return bad("slot out of range");
        const KdNode& h = H[it.host];
        const KdNode& d = S[it.slot];
        if (is_leaf(h)) {
            if (!(d.w1 & kDevLeafBit)) return bad("host leaf is a device interior node");
            const uint32_t n = h.w1 >> 1, kind = (d.w1 >> 29) & 3u, pay = d.w1 & kDevLeafPayload;
            if (kind != (n >= 3u ? 3u : n)) return bad("leaf kind");
            if (n == 1u && d.w0 != h.w0) return bad("single prim id");
            if (n == 2u && (d.w0 != T.leaf_prim_ids[h.w0] || pay != T.leaf_prim_ids[h.w0 + 1])) return bad("two-prim ids");
            if (n >= 3u) {
                if (pay != n) return bad("leaf size");
                if (d.w0 & 1u) return bad("id list is not 8-byte aligned");
                if ((size_t)d.w0 + ((n + 1u) & ~1u) > n_ids) return bad("id list out of range");
                for (uint32_t i = 0; i < n; ++i)
                    if (ids[d.w0 + i] != T.leaf_prim_ids[h.w0 + i]) return bad("id list contents");
                // a list inside the node buffer must not straddle into another line's nodes: it is
                // either wholly inside one 128-byte line or in the overflow area (checked by contents)
            }
        } else {
            if (d.w1 & kDevLeafBit) return bad("host interior node is a device leaf");
            if (d.w0 != h.w0) return bad("split position");
            if ((d.w1 & 3u) != ((h.w1 >> 1) & 3u)) return bad("axis");
            const uint32_t pair = d.w1 >> 2;
            if (pair & 1u) return bad("child pair is not 16-byte aligned");
            st.push_back(It{h.w1 >> 3, pair + 1u});
            st.push_back(It{it.host + 1u, pair});
        }
    }
    return seen == H.size() ? true : bad("node count");
}

}  // namespace gigl
