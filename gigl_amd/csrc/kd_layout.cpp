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
        // unless nearly all of the current line is still free
        // measured with tools/layout_sim on the bench batch: distinct node lines per ray 18.96 / 17.24 /
        // 15.91 / 15.00 / 14.82 / 15.80 for 4 / 8 / 16 / 24 / 28 / 32 dwords
        constexpr uint32_t min_free = 28;
        if (sub[r.host] <= kLine ? sub[r.host] > free_dw : (free_dw < min_free || first_cost > free_dw)) new_line();
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

// Triple layout (see kd_layout.h).  Lines of 8 units (16 B each): records grow from the front of a
// line in blocks of 2 or 4 (the grandchildren of one record), leaf records and id arrays from the
// back; what does not fit beside its record goes to lines of its own at the end of the buffer.
bool layout_triples(const KdTreeHost& T, DeviceTree& dt) {
    constexpr uint32_t kLineU = 8;
    dt.ids_inline = true;
    dt.triples = true;
    dt.slots.clear();
    dt.ids.clear();
    dt.block_ids.clear();
    const uint32_t block_min = dt.leaf_block_min;
    const std::vector<KdNode>& H = T.nodes;
    if (dt.slot_of_host) dt.slot_of_host->assign(H.size(), 0u);
    if (dt.list_dw_of_host) dt.list_dw_of_host->assign(H.size(), ~0u);
    std::vector<uint32_t> W(4 * kLineU, 0u);
    bool ok = true;
    auto fill_pad = [&](uint32_t first_unit, uint32_t n_units) {   // unused units read as empty leaf records
        for (uint32_t u = first_unit; u < first_unit + n_units; ++u) W[4 * u + 3] = 3u;
    };
    fill_pad(0, kLineU);
    auto finish = [&]() {
        W.resize(W.size() + 4, 0u);
        dt.slots.resize(W.size() / 2);
        std::memcpy(dt.slots.data(), W.data(), W.size() * 4);
        return ok;
    };
    auto n_of = [&](uint32_t k) { return H[k].w1 >> 1; };
    auto id_of = [&](uint32_t k, uint32_t i) { return n_of(k) == 1u ? H[k].w0 : T.leaf_prim_ids[H[k].w0 + i]; };
    auto ids_units = [](uint32_t n) { return (n + 3u) / 4u; };
    uint32_t line = 0, front = 2, back = kLineU;   // current line (unit index of its start), unit offsets in it
    uint32_t ov_line = 0, ov_used = kLineU;           // overflow: shared lines filled front to back
    auto alloc_aux = [&](uint32_t nu) -> uint32_t {   // from the back of the current line, else the overflow lines
        if (nu <= back - front) { back -= nu; return line + back; }
        if (nu <= kLineU - ov_used) { const uint32_t u = ov_line + ov_used; ov_used += nu; return u; }
        const uint32_t u = (uint32_t)(W.size() / 4);
        const uint32_t nl = (nu + kLineU - 1) / kLineU;
        W.resize(W.size() + (size_t)4 * kLineU * nl, 0u);
        fill_pad(u, nl * kLineU);
        if (nu % kLineU) { ov_line = u + (nl - 1) * kLineU; ov_used = nu % kLineU; }
        return u;
    };
    auto write_ids = [&](uint32_t u, uint32_t k) {
        const uint32_t n = n_of(k);
        for (uint32_t i = 0; i < n; ++i) W[4 * u + i] = id_of(k, i);
        for (uint32_t i = n; i < 4 * ids_units(n); ++i) W[4 * u + i] = 0u;
    };
    auto in_block = [&](uint32_t k) { return block_min > 0u && n_of(k) >= block_min; };
    auto write_leaf_record = [&](uint32_t u, uint32_t k) {
        const uint32_t n = n_of(k);
        if (n >= (1u << 29)) ok = false;
        W[4 * u] = W[4 * u + 1] = W[4 * u + 2] = 0u;
        W[4 * u + 3] = 3u | (n << 2);
        if (in_block(k)) {   // the leaf's triangles, copied in list order into lines of their own
            const size_t first_line = dt.block_ids.size() / 3u;
            if (first_line > 0xFFFFFFFFull) ok = false;
            W[4 * u] = (uint32_t)first_line;
            W[4 * u + 3] |= kLeafBlockFlag;
            for (uint32_t i = 0; i < n; ++i) dt.block_ids.push_back(id_of(k, i));
            while (dt.block_ids.size() % 3u) dt.block_ids.push_back(kLeafBlockPad);
        } else if (n <= 3u) {
            for (uint32_t i = 0; i < n; ++i) W[4 * u + i] = id_of(k, i);
        } else {
            const uint32_t a = alloc_aux(ids_units(n));
            W[4 * u] = a;
            write_ids(a, k);
            if (dt.list_dw_of_host) (*dt.list_dw_of_host)[k] = 4 * a;
        }
    };
    auto leaf_payload = [&](uint32_t k) -> uint32_t {   // a leaf at a child position
        const uint32_t n = n_of(k);
        if (n == 0u) return 0u;
        if (n == 1u) { if (H[k].w0 >= (1u << 30)) ok = false; return (H[k].w0 << 2) | 1u; }
        if (in_block(k) && n <= kPayloadBlockMaxN && dt.block_ids.size() / 3u < (1u << kPayloadBlockLineBits)) {
            // the block is named by the payload itself: no leaf record, one dependent load less
            const uint32_t first_line = (uint32_t)(dt.block_ids.size() / 3u);
            for (uint32_t i = 0; i < n; ++i) dt.block_ids.push_back(id_of(k, i));
            while (dt.block_ids.size() % 3u) dt.block_ids.push_back(kLeafBlockPad);
            return (first_line << 7) | (n << 2) | 3u;
        }
        const uint32_t u = alloc_aux(1);
        if (u >= (1u << 30)) ok = false;
        write_leaf_record(u, k);
        if (dt.list_dw_of_host && n <= 3u) (*dt.list_dw_of_host)[k] = 4 * u;
        return (u << 2) | 2u;
    };
    // record of node k at unit u; returns true if k is interior and has grandchildren to place
    auto place_record = [&](uint32_t u, uint32_t k) -> bool {
        if (dt.slot_of_host) (*dt.slot_of_host)[k] = 2 * u;
        if (is_leaf(H[k])) { write_leaf_record(u, k); return false; }
        const uint32_t kids[2] = {k + 1u, H[k].w1 >> 3};
        uint32_t meta = (H[k].w1 >> 1) & 3u;
        W[4 * u] = H[k].w0;
        bool more = false;
        for (int j = 0; j < 2; ++j) {
            const uint32_t c = kids[j];
            if (dt.slot_of_host) (*dt.slot_of_host)[c] = 2 * u;
            if (is_leaf(H[c])) {
                W[4 * u + 1 + j] = leaf_payload(c);
                meta |= 3u << (2 + 2 * j);
            } else {
                W[4 * u + 1 + j] = H[c].w0;
                meta |= ((H[c].w1 >> 1) & 3u) << (2 + 2 * j);
                more = true;
            }
        }
        W[4 * u + 3] = meta;
        return more;
    };
    if (H.empty()) return finish();
    if (is_leaf(H[0])) { place_record(0, 0); return finish(); }

    const uint32_t n_nodes = (uint32_t)H.size();
    auto arr_units = [&](uint32_t k) { return n_of(k) > 3u && !in_block(k) ? ids_units(n_of(k)) : 0u; };
    auto aux_of = [&](uint32_t g) -> uint32_t {   // units a record of node g wants beside it
        if (is_leaf(H[g])) return arr_units(g);
        uint32_t a = 0;
        const uint32_t kids[2] = {g + 1u, H[g].w1 >> 3};
        for (uint32_t c : kids)
            if (is_leaf(H[c]) && n_of(c) >= 2u && !(in_block(c) && n_of(c) <= kPayloadBlockMaxN)) a += 1u + arr_units(c);
        return a;
    };
    auto block_of = [&](uint32_t k, uint32_t g[4]) -> uint32_t {   // the grandchildren of interior node k
        uint32_t n = 0;
        const uint32_t kids[2] = {k + 1u, H[k].w1 >> 3};
        for (uint32_t c : kids)
            if (!is_leaf(H[c])) { g[n++] = c + 1u; g[n++] = H[c].w1 >> 3; }
        return n;
    };
    // units of everything below an interior record node (saturating); only filled at record positions
    std::vector<uint32_t> sub(n_nodes, 0u);
    {
        std::vector<uint8_t> is_rec(n_nodes, 0);
        is_rec[0] = 1;
        for (uint32_t k = 0; k < n_nodes; ++k) {
            if (!is_rec[k] || is_leaf(H[k])) continue;
            uint32_t g[4];
            const uint32_t n = block_of(k, g);
            for (uint32_t i = 0; i < n; ++i) is_rec[g[i]] = 1;
        }
        for (uint32_t k = n_nodes; k-- > 0;) {
            if (!is_rec[k] || is_leaf(H[k])) continue;
            uint32_t g[4];
            const uint32_t n = block_of(k, g);
            uint64_t s = n;
            for (uint32_t i = 0; i < n; ++i) s += aux_of(g[i]) + sub[g[i]];
            sub[k] = s > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)s;
        }
    }
    struct Cand { float sa; uint32_t host; uint32_t unit; float b[6]; };
    struct BySa { bool operator()(const Cand& a, const Cand& b) const { return a.sa < b.sa; } };
    auto area = [](const float* b) {
        const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
        return dx * dy + dy * dz + dz * dx;
    };
    auto clip = [&](float* b, uint32_t parent, int j) {   // cell of child j of interior node `parent`
        const unsigned axis = (H[parent].w1 >> 1) & 3u;
        float split;
        std::memcpy(&split, &H[parent].w0, 4);
        if (j == 0) b[3 + axis] = split; else b[axis] = split;
    };
    std::vector<Cand> roots, deferred;
    std::priority_queue<Cand, std::vector<Cand>, BySa> heap;
    {
        Cand c;
        c.host = 0; c.unit = 0;
        for (int a = 0; a < 3; ++a) { c.b[a] = T.box_min[a]; c.b[3 + a] = T.box_max[a]; }
        c.sa = area(c.b);
        if (place_record(0, 0)) roots.push_back(c);
    }
    auto new_line = [&]() {
        line = (uint32_t)(W.size() / 4);
        W.resize(W.size() + (size_t)4 * kLineU, 0u);
        fill_pad(line, kLineU);
        front = 0; back = kLineU;
    };
    while (!roots.empty()) {
        const Cand r = roots.back();
        roots.pop_back();
        uint32_t g[4];
        const uint32_t free_u = back - front;
        constexpr uint32_t min_free_u = 6;   // tools/layout_sim: 18.50 / 18.38 / 17.63 / 18.08 lines per ray for 2 / 4 / 6 / 8
        if (sub[r.host] <= kLineU ? sub[r.host] > free_u : (free_u < min_free_u || block_of(r.host, g) > free_u)) new_line();
        heap.push(r);
        deferred.clear();
        while (!heap.empty()) {
            const Cand c = heap.top();
            heap.pop();
            const uint32_t n = block_of(c.host, g);
            if (n > back - front) { deferred.push_back(c); continue; }
            const uint32_t at = line + front;
            front += n;
            if ((at & 1u) || (at >> 1) >= (1u << kTriplePtrBits)) ok = false;
            W[4 * c.unit + 3] |= (at >> 1) << 6;
            // which child of c.host each grandchild hangs under, for the cell boxes
            const uint32_t kids[2] = {c.host + 1u, H[c.host].w1 >> 3};
            uint32_t idx = 0;
            for (int j = 0; j < 2; ++j) {
                if (is_leaf(H[kids[j]])) continue;
                for (int jj = 0; jj < 2; ++jj, ++idx) {
                    Cand nx;
                    nx.host = g[idx]; nx.unit = at + idx;
                    std::memcpy(nx.b, c.b, sizeof(nx.b));
                    clip(nx.b, c.host, j);
                    clip(nx.b, kids[j], jj);
                    nx.sa = area(nx.b);
                    if (place_record(nx.unit, nx.host)) heap.push(nx);
                }
            }
        }
        std::stable_sort(deferred.begin(), deferred.end(), [&](const Cand& a, const Cand& b) {
            const bool sa_ = sub[a.host] <= kLineU, sb_ = sub[b.host] <= kLineU;
            if (sa_ != sb_) return sa_;
            return a.sa > b.sa;
        });
        for (size_t i = deferred.size(); i-- > 0;) roots.push_back(deferred[i]);
    }
    return finish();
}

namespace {

bool verify_triples(const KdTreeHost& T, const DeviceTree& D, const char** what) {
    auto bad = [&](const char* m) { if (what) *what = m; return false; };
    const std::vector<KdNode>& H = T.nodes;
    const uint32_t* W = reinterpret_cast<const uint32_t*>(D.slots.data());
    const size_t n_units = D.slots.size() / 2;
    if (n_units < 1) return bad("no root record");
    auto n_of = [&](uint32_t k) { return H[k].w1 >> 1; };
    auto id_of = [&](uint32_t k, uint32_t i) { return n_of(k) == 1u ? H[k].w0 : T.leaf_prim_ids[H[k].w0 + i]; };
    auto leaf_record_ok = [&](uint32_t u, uint32_t k) -> bool {
        if (u >= n_units) return false;
        const uint32_t n = n_of(k);
        if (D.leaf_block_min > 0u && n >= D.leaf_block_min) {
            if (W[4 * u + 3] != (3u | (n << 2) | kLeafBlockFlag)) return false;
            const size_t first = (size_t)W[4 * u] * 3u;
            if (first + n > D.block_ids.size()) return false;
            for (uint32_t i = 0; i < n; ++i) if (D.block_ids[first + i] != id_of(k, i)) return false;
            for (size_t i = first + n; i % 3u; ++i) if (D.block_ids[i] != kLeafBlockPad) return false;
            return true;
        }
        if (W[4 * u + 3] != (3u | (n << 2))) return false;
        if (n <= 3u) {
            for (uint32_t i = 0; i < n; ++i) if (W[4 * u + i] != id_of(k, i)) return false;
            return true;
        }
        const uint32_t a = W[4 * u];
        if ((size_t)a + (n + 3u) / 4u > n_units) return false;
        for (uint32_t i = 0; i < n; ++i) if (W[4 * a + i] != id_of(k, i)) return false;
        return true;
    };
    if (H.empty()) return W[3] == 3u ? true : bad("empty tree must be an empty leaf record");
    struct It { uint32_t host, unit; };
    std::vector<It> st(1, It{0u, 0u});
    size_t seen = 0;
    while (!st.empty()) {
        const It it = st.back();
        st.pop_back();
        ++seen;
        if (it.unit >= n_units) return bad("record out of range");
        const KdNode& h = H[it.host];
        const uint32_t* r = W + 4 * (size_t)it.unit;
        if (is_leaf(h)) {
            if (!leaf_record_ok(it.unit, it.host)) return bad("leaf record");
            continue;
        }
        if ((r[3] & 3u) != ((h.w1 >> 1) & 3u)) return bad("axis of a record");
        if (r[0] != h.w0) return bad("split of a record");
        const uint32_t kids[2] = {it.host + 1u, h.w1 >> 3};
        uint32_t next = (r[3] >> 6) * 2u;
        for (int j = 0; j < 2; ++j) {
            const KdNode& c = H[kids[j]];
            const uint32_t type = (r[3] >> (2 + 2 * j)) & 3u, slot = r[1 + j];
            ++seen;
            if (is_leaf(c)) {
                if (type != 3u) return bad("child leaf type");
                const uint32_t n = c.w1 >> 1, kind = slot & 3u;
                if (kind == 3u) {   // a triangle block named by the payload
                    const size_t first = (size_t)(slot >> 7) * 3u;
                    if (D.leaf_block_min == 0u || n < D.leaf_block_min || n > kPayloadBlockMaxN || ((slot >> 2) & 31u) != n ||
                        first + n > D.block_ids.size())
                        return bad("child leaf block payload");
                    for (uint32_t i = 0; i < n; ++i) if (D.block_ids[first + i] != id_of(kids[j], i)) return bad("child leaf block ids");
                    for (size_t i = first + n; i % 3u; ++i) if (D.block_ids[i] != kLeafBlockPad) return bad("child leaf block padding");
                } else if (n == 0u ? slot != 0u : n == 1u ? (kind != 1u || (slot >> 2) != c.w0)
                                                          : (kind != 2u || !leaf_record_ok(slot >> 2, kids[j])))
                    return bad("child leaf payload");
            } else {
                if (type != ((c.w1 >> 1) & 3u)) return bad("child axis");
                if (slot != c.w0) return bad("child split");
                st.push_back(It{c.w1 >> 3, next + 1u});
                st.push_back(It{kids[j] + 1u, next});
                next += 2u;
            }
        }
    }
    return seen == H.size() ? true : bad("node count");
}

}  // namespace

bool layout_verify(const KdTreeHost& T, const DeviceTree& D, const char** what) {
    if (D.triples) return verify_triples(T, D, what);
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
