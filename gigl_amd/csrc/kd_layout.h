// Device layout of the k-d tree (product code, host side).
//
// The tree built by kd_build() keeps the reference's encoding and depth-first numbering
// (/root/reference/Source/RT/KdTree.h:22-65, KdTree.hpp:254-262).  The traversal kernel reads a
// re-addressed copy: same topology, same leaf contents in the same order, different addresses.
// Three layouts are provided.  The level and clustered layouts use the pair-addressed node encoding
// read by trace_kernel<TRIPLES = false> (the triple layout, TRIPLES = true, is described further down):
//   interior: w0 = split_pos, w1 = slot of the children pair << 2 | axis (bottom child first)
//   leaf:     w1 bit 31 = 1, bits 29-30 = kind, bits 0-28 = payload
//             kind 0 empty | 1 one prim (w0 = id) | 2 two prims (w0 = id0, payload = id1)
//             kind 3: n >= 3 prims, payload = n, w0 = index of the id list in 4-byte units
//                     (into DeviceTree::ids for the level layout, into the node buffer itself for
//                     the clustered layout, which keeps the list in the leaf's own cache line).
//                     Lists start on an 8-byte boundary and are padded to an even length, so
//                     the kernel reads two ids with one load.
#pragma once
#include <cstdint>
#include <vector>

#include "kd_build.h"

namespace gigl {

constexpr uint32_t kDevLeafBit = 0x80000000u, kDevLeafPayload = 0x1FFFFFFFu;

// Triple layout (layout_triples): 16-byte records, one per node at an even depth below the root.
//   interior record of node N with children A (bottom), B (top):
//     x = split_N, y = slot of A, z = slot of B (split if interior, else leaf payload)
//     w = axis_N | type_A << 2 | type_B << 4 | ptr << 6      (type: 0-2 axis, 3 leaf)
//     ptr * 2 = unit (16 B) index of the block holding the records of A's children (if A is
//     interior) followed by those of B's children (if B is interior)
//   leaf payload in a slot: kind (bits 0-1) 0 empty | 1 one prim, id = payload >> 2
//                           | 2 list, leaf record unit = payload >> 2
//                           | 3 triangle block named by the payload itself (leaf_block_min <= n <= 31):
//                             n = bits 2-6, first 128-byte line of the block = payload >> 7 (no leaf record)
//   leaf record (a leaf at an even depth, or the list of a child-position leaf):
//     w = 3 | n << 2;  n <= 3: x, y, z = ids;  n > 3: x = unit index of the id array (dwords,
//     padded to an even count, read 8 bytes at a time)
//   leaf record with a triangle block (DeviceTree::leaf_block_min > 0 and n >= it):
//     w = 3 | n << 2 | 1 << 31;  x = index of the first 128-byte line of the leaf's private block of
//     triangle copies in a buffer of its own (three 40-byte records per line: p0, e0, e1, id), which
//     the kernel reads in order instead of an id list plus one scattered 64-byte record per id.
//     DeviceTree::block_ids lists the ids, three per line, padded with 0xFFFFFFFF.
// One 16-byte load serves two levels of the descent.
constexpr uint32_t kTriplePtrBits = 26;
constexpr uint32_t kLeafBlockFlag = 0x80000000u, kLeafBlockPad = 0xFFFFFFFFu;
constexpr uint32_t kPayloadBlockMaxN = 31, kPayloadBlockLineBits = 25;

struct DeviceTree {
    std::vector<KdNode> slots;          // 8-byte slots: nodes (and, clustered layout, inline id lists)
    bool ids_inline = false;            // kind-3 leaves index `slots` (as uint32) instead of `ids`
    std::vector<uint32_t> ids;          // level layout: the id lists, each padded to an even length
    // optional diagnostics (filled when non-null on entry): device slot of every host node
    std::vector<uint32_t>* slot_of_host = nullptr;
    // triple layout diagnostics: dword address of the extra record / id array a leaf reads (~0: none)
    std::vector<uint32_t>* list_dw_of_host = nullptr;
    bool triples = false;               // slots hold the triple layout (two slots per record)
    // triple layout, in: leaves with at least this many prims get a private triangle block (0: none);
    // out: the ids of the blocks' records, three per 128-byte line
    uint32_t leaf_block_min = 0;
    std::vector<uint32_t> block_ids;
};

// Level layout: treelets of `levels` breadth-first levels of child pairs that never straddle a
// 128-byte line (round-1 layout).
void layout_levels(const KdTreeHost& T, int levels, DeviceTree& out);

// Clustered layout: every 128-byte line is filled greedily from a cluster root with the child pairs
// most likely to be read next (largest cell surface area first, the probability a random line that
// crosses the parent's cell crosses the child's), together with the id lists of the leaves it
// holds.  Subtrees that fit in what is left of a line are kept whole.
void layout_clustered(const KdTreeHost& T, DeviceTree& out);

// Triple layout, clustered into 128-byte lines like layout_clustered.  Returns false if the tree
// does not fit the encoding (more than 2 GB of records, ids beyond 2^30); the caller falls back.
bool layout_triples(const KdTreeHost& T, DeviceTree& out);

// Walks the host tree and the device tree side by side: same splits, axes and child order, and at
// every leaf the same primitives in the same order.  Returns false and a message at the first
// difference (used by the CPU tests through gigl_debug_check_layout).
bool layout_verify(const KdTreeHost& T, const DeviceTree& D, const char** what);

}  // namespace gigl
