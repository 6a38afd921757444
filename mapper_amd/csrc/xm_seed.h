// xmapper-hip device core: read-side HashBlock pyramid, index probes, adaptive seed walk and voting.
// Replaces (per read, on the GPU):
//   M/HashBlock.java, M/HashBlock_BaseRow.java, M/HashBlock_ParentRow.java, M/HashBlock_Stream.java, M/HashBlock_Pyramid.java
//   M/PackedMap.java:160-236 (get / getNumMatchesLowerBound), M/Readable_HashBlock_Database.java:22-90
//   M/HashBlockPath.java, M/Counting_HashBlockPath.java, M/HashBlockMatch_Counter.java, M/HashBlockPaths_Counter.java,
//   M/SequenceMatch.java, M/QueryMatch.java
// Design: a pyramid level is a pure function of the level below (block at p = merge(block at p, its successor)), so
// levels are materialised whole, on demand, as sorted arrays; Java's TreeMaps/HashMaps become small flat arrays.
#pragma once
#include "xm_defs.h"

// which functions of the index walk are out of line (experiment define XM_WALK_INLINE: 1 the path's two, 2 + the component's block fetch, 3 + the vote update, 4 = 2 + the component's step itself)
#ifndef XM_WALK_INLINE
#define XM_WALK_INLINE 2  // (light pass 22.4-23.2 -> 21.2-21.3 ms per 1 M reads: a call of pathAdvance saved and restored 24 registers, 30 times a read; 4: the same; 3: 21.3)
#endif
#if XM_WALK_INLINE >= 1
#define XM_NOINL_W1 XM_INL
#else
#define XM_NOINL_W1 XM_NOINL
#endif
#if XM_WALK_INLINE >= 2
#define XM_NOINL_W2 XM_INL
#else
#define XM_NOINL_W2 XM_NOINL
#endif
#if XM_WALK_INLINE == 3
#define XM_NOINL_W3 XM_INL
#else
#define XM_NOINL_W3 XM_NOINL
#endif
#if XM_WALK_INLINE >= 4
#define XM_NOINL_W4 XM_INL
#else
#define XM_NOINL_W4 XM_NOINL
#endif

namespace xm {

// ---------------------------------------------------------------- pyramid
struct alignas(16) PBlock {  // 16 bytes, moved as one 16-byte word
  uint16_t start, len;
  int32_t fwd, rev;
  uint8_t flags;  // 1 requestMergeLeft, 2 requestMergeRight, 4 nextRequestMergeLeft, 8 nextRequestMergeRight
  int8_t gapDir;
  int16_t extraGap;
};
enum : uint8_t { F_RML = 1, F_RMR = 2, F_NRML = 4, F_NRMR = 8 };
// one 16-byte memory operation per block instead of one per field group (each lane's pyramid is in its own arena, so every
// memory instruction of a wave touches 64 different lines: the instruction count is what the memory pipeline sees)
struct alignas(16) PBlockWord { uint32_t w[4]; };
XM_INL PBlock pbUnpack(const PBlockWord& v) {
  PBlock b;
  b.start = (uint16_t)(v.w[0] & 0xFFFFu); b.len = (uint16_t)(v.w[0] >> 16); b.fwd = (int32_t)v.w[1]; b.rev = (int32_t)v.w[2];
  b.flags = (uint8_t)(v.w[3] & 0xFFu); b.gapDir = (int8_t)((v.w[3] >> 8) & 0xFFu); b.extraGap = (int16_t)(v.w[3] >> 16);
  return b;
}
XM_INL PBlockWord pbPack(const PBlock& b) {
  PBlockWord v;
  v.w[0] = (uint32_t)b.start | ((uint32_t)b.len << 16); v.w[1] = (uint32_t)b.fwd; v.w[2] = (uint32_t)b.rev;
  v.w[3] = (uint32_t)b.flags | ((uint32_t)(uint8_t)b.gapDir << 8) | ((uint32_t)(uint16_t)b.extraGap << 16);
  return v;
}

XM_INL int maxGapmerNumBasepairsUsed(int startingLength) { return startingLength + startingLength * 9 / 8 + 1; }  // M/HashBlock.java:11-13

XM_INL PBlock level0Block(uint8_t code, int index) {  // M/HashBlock.java:60-65,171-188
  PBlock b;
  int f = (code == 1) ? 0 : (code == 2) ? 1 : (code == 4) ? 2 : 3;
  b.start = (uint16_t)index;
  b.len = 1;
  b.fwd = f;
  b.rev = 3 - f;
  b.flags = (uint8_t)(((f / 2 == 0) ? F_RML : F_RMR) | ((f % 2 == 0) ? F_NRML : F_NRMR));
  b.gapDir = 0;
  b.extraGap = 0;
  return b;
}

XM_INL int32_t mergeHash(int leftLength, int32_t leftHash, int rightLength, int32_t rightHash) {  // M/HashBlock.java:261-269
  int64_t rotatedLeft = ((int64_t)leftHash + 1) * (54323 + 323 * (int64_t)rightLength);
  int64_t rotatedRight = (int64_t)jadd(rightHash, 1) * (int64_t)leftLength;
  uint64_t top = (uint64_t)rotatedLeft + (uint64_t)rotatedRight;
  return jadd((int32_t)(uint32_t)top, (int32_t)(uint32_t)((uint64_t)((int64_t)top >> 32)));
}

template <typename B>
XM_INL bool shouldMergeBlocks(const B& l, const B& r) {  // M/HashBlock_ParentRow.java:200-208
  if ((int)l.start + l.len < (int)r.start) return false;
  return (l.flags & F_RMR) || (r.flags & F_RML);
}

// The merge flags and the gap direction of a merged block (M/HashBlock.java:192-259) depend on five small facts only: which parent
// anchors (L.fwd vs R.rev), the order of the merged hashes, the two "next" flags of either parent and the order of the parents'
// lengths.  The rule is evaluated once per combination at compile time; a merge then costs two hashes and one table read (the
// branchy rule itself was ~3/4 of the instructions of building a read's pyramid).
//   anchor: 0 none, 1 left parent, 2 right parent;  fr: 0 fwd<rev, 1 equal, 2 fwd>rev;  aBits/oBits: bit0 nextRequestMergeLeft,
//   bit1 nextRequestMergeRight of the anchor / the other parent;  lc: 0 L.len<R.len, 1 equal, 2 L.len>R.len
//   result: bits 0-3 the four flags, bits 4-5 gapDirection + 1
constexpr uint8_t mergeRule(int anchor, int fr, int aBits, int oBits, int lc) {
  bool rml = true, rmr = true, nrml = true, nrmr = true;
  if (anchor != 0 && fr != 1) {
    bool anchorIsRight = anchor == 2;
    bool isReverse = fr == 0;
    bool invert = isReverse == anchorIsRight;
    bool aL = (aBits & 1) != 0, aR = (aBits & 2) != 0;
    if (aL && aR) { if (anchorIsRight) aR = false; else aL = false; }
    bool oL = (oBits & 1) != 0, oR = (oBits & 2) != 0;
    if (oL && oR) { if (!anchorIsRight) oL = false; else oR = false; }  // otherParent == rightParent <=> anchor is left
    rml = aL != invert;
    rmr = aR != invert;
    nrml = oL != invert;
    nrmr = oR != invert;
  }
  if (lc != 1) {
    rml = lc == 2;
    rmr = !rml;
    nrml = !rml;
    nrmr = !nrml;
  }
  if (fr != 1) {
    if (rml && rmr) { rml = fr == 2; rmr = !rml; }
    if (nrml && nrmr) { nrml = rml; nrmr = !nrml; }
  }
  int gd = 0;
  if (rml != rmr) gd = rml ? 1 : -1;
  else if (anchor != 0) gd = (anchor == 2) ? 1 : -1;
  return (uint8_t)((rml ? F_RML : 0) | (rmr ? F_RMR : 0) | (nrml ? F_NRML : 0) | (nrmr ? F_NRMR : 0) | ((gd + 1) << 4));
}
struct MergeRuleTable { uint8_t v[3 * 3 * 4 * 4 * 3]; };
constexpr MergeRuleTable makeMergeRuleTable() {
  MergeRuleTable t{};
  for (int anchor = 0; anchor < 3; anchor++) for (int fr = 0; fr < 3; fr++) for (int a = 0; a < 4; a++) for (int o = 0; o < 4; o++) for (int lc = 0; lc < 3; lc++)
    t.v[(((anchor * 3 + fr) * 4 + a) * 4 + o) * 3 + lc] = mergeRule(anchor, fr, a, o, lc);
  return t;
}
static constexpr MergeRuleTable kMergeRuleHost = makeMergeRuleTable();
#if defined(__HIPCC__)
static __device__ const MergeRuleTable kMergeRuleDev = makeMergeRuleTable();
#endif
#if defined(__HIP_DEVICE_COMPILE__)
// the kernel copies the table into LDS before anything else (xmLoadMergeRule): a table read is then ~64 cycles, next to the hashes
__shared__ uint8_t xm_merge_rule_lds[sizeof(MergeRuleTable)];
XM_INL void xmLoadMergeRule() {
  for (int i = (int)threadIdx.x; i < (int)sizeof(MergeRuleTable); i += (int)blockDim.x) xm_merge_rule_lds[i] = kMergeRuleDev.v[i];
  __syncthreads();
}
XM_INL uint8_t mergeRuleLookup(int idx) { return xm_merge_rule_lds[idx]; }
#else
XM_INL void xmLoadMergeRule() {}
XM_INL uint8_t mergeRuleLookup(int idx) { return kMergeRuleHost.v[idx]; }
#endif

template <typename B>
XM_INL B mergeBlocks(const B& L, const B& R) {  // M/HashBlock.java:20-44,192-259
  B b;
  int start = L.start, len = (int)R.start + R.len - (int)L.start;
  b.start = (decltype(b.start))start;
  b.len = (decltype(b.len))len;
  b.fwd = mergeHash(L.len, L.fwd, R.len, R.fwd);
  b.rev = mergeHash(R.len, R.rev, L.len, L.rev);
  const int anchor = (L.fwd != R.rev) ? ((L.fwd > R.rev) ? 2 : 1) : 0;  // 0 none, 1 left parent, 2 right parent
  const int fr = (b.fwd < b.rev) ? 0 : ((b.fwd == b.rev) ? 1 : 2);
  const int lc = (L.len < R.len) ? 0 : ((L.len == R.len) ? 1 : 2);
  const int lBits = (L.flags >> 2) & 3, rBits = (R.flags >> 2) & 3;  // F_NRML = 4, F_NRMR = 8
  const int aBits = (anchor == 2) ? rBits : lBits, oBits = (anchor == 2) ? lBits : rBits;
  const uint8_t rule = mergeRuleLookup((((anchor * 3 + fr) * 4 + aBits) * 4 + oBits) * 3 + lc);
  b.flags = (uint8_t)(rule & 15);
  b.gapDir = (int8_t)((int)(rule >> 4) - 1);
  b.extraGap = (int16_t)(((int)L.len + (int)R.len - len) / 4);
  return b;
}

// a block handed to the database: a gapmer (XX_X) or, when gapDirection == 0 / gapmers are off, the block itself
struct QBlock {
  int32_t start, len, used, fwd, rev;
  int32_t id;     // identity of the Java object
  uint8_t flags;  // merge flags (all clear for a Gapped_HashBlock, M/Gapped_HashBlock.java:7-11)
  XM_INL int end() const { return start + len; }
  XM_INL bool isPrimaryPolarity() const {  // M/HashBlock.java:329-334
    bool rml = (flags & F_RML) != 0, rmr = (flags & F_RMR) != 0;
    if (rml != rmr) return rml;
    return fwd >= rev;
  }
  XM_INL int32_t lookupKey() const { return isPrimaryPolarity() ? fwd : rev; }
};

XM_INL int gapmerCharCode(uint8_t code) {  // M/HashBlock.java:152-169 on Basepairs.decode(code)
  return code == 1 ? 1 : code == 2 ? 2 : code == 4 ? 3 : code == 8 ? 4 : 0;
}

// M/HashBlock.java:67-150.  0 = null, 1 = block itself, 2 = new gapmer
template <typename B>
XM_INL int withGapAndExtension(const B& b, const SeqView& seq, QBlock& out) {
  int targetExtraLength = b.len;
  int32_t mx = b.fwd > b.rev ? b.fwd : b.rev;
  targetExtraLength += jabs(mx) % 3;
  targetExtraLength += b.extraGap;
  int gapLength = b.len / 2;
  int extensionLength = targetExtraLength - gapLength;
  out.flags = 0;
  if (b.gapDir == 0) {
    out.start = b.start; out.len = b.len; out.used = b.len; out.fwd = b.fwd; out.rev = b.rev; out.flags = b.flags;
    return 1;
  }
  int32_t extensionHash = 0;
  if (b.gapDir < 0) {
    int extensionEnd = (int)b.start - gapLength;
    int extensionStart = extensionEnd - extensionLength;
    if (extensionStart < 0) return 0;
    for (int i = extensionEnd - 1; i >= extensionStart; i--) {
      extensionHash = jmul(extensionHash, 7654337);
      extensionHash = jadd(extensionHash, gapmerCharCode(seq.at(i)));
    }
    out.start = extensionStart;
    out.len = extensionLength + gapLength + b.len;
  } else {
    int extensionStart = (int)b.start + b.len + gapLength;
    int extensionEnd = extensionStart + extensionLength;
    if (extensionEnd > seq.len) return 0;
    for (int i = extensionStart; i < extensionEnd; i++) {
      extensionHash = jmul(extensionHash, 7654337);
      extensionHash = jadd(extensionHash, gapmerCharCode(bpComplement(seq.at(i))));
    }
    out.start = b.start;
    out.len = b.len + gapLength + extensionLength;
  }
  out.fwd = jadd(b.fwd, extensionHash);
  out.rev = jadd(b.rev, extensionHash);
  out.used = b.len + extensionLength;
  return 2;
}

// ---------------------------------------------------------------- reads with ambiguous bases (M/MultiHashBlock.java, M/ConditionalHashBlock.java,
// M/SequenceCondition.java, the multi branch of M/HashBlock_ParentRow.java:69-191, M/HashBlock_BaseRow.java:20-49)
// A block that covers an ambiguous base is a list of possibilities, each either a HashBlock or "no block", under a condition
// (position -> base).  The walk never probes such a block (HashBlockPath.skipMultiblocks) but it has to know where they are, and
// whether one exists depends on the full expansion, so the rule is restated as written.  Only reads that contain a non-ACGT base
// take this path.  Like the reference, it bounds the combinations per block (64) and nothing else: a mate may consist of ambiguous bases
// only.  Everything it keeps - possibilities, conditions, the expansion's stack - is in pools of the read's arena whose capacities follow
// the scratch scale, so a read that outgrows them is run again with more scratch like any other (XM_ST_OVERFLOW).
constexpr uint8_t F_MULTI = 0x80;   // PBlock::flags of a stored multi block: fwd = first possibility in the pool, rev = their number
constexpr int XM_MAX_COMBINATIONS = 64;  // HashBlock_ParentRow.maxNumCombinationsToExpand
// SequenceCondition: (position -> base) constraints, ascending by position.  Its only operation is intersect (:27-106): null if the two
// disagree on a position, else their union (the reference returns an operand when it contains the other, which is the union too).
// An entry is (position << 2) | base with A C G T = 0 1 2 3; a condition is a run of entries in a pool.
typedef uint32_t CondEnt;
struct Poss {  // ConditionalHashBlock
  PBlock block;
  int32_t condOff, condLen;  // MultiStore::conds[condOff .. condOff + condLen)
  int32_t hasBlock, pad;
};
struct PossView {  // a possibility as expand() reads it: of a stored multi block, of an ambiguous base (its one entry is `own`) or of a single block (no condition)
  PBlock block;
  const CondEnt* cond; int32_t condLen;
  int32_t hasBlock;
  CondEnt own;
};
// a.intersect(b) into out[0 .. cap): the length of the union, -1 = conflict (Java null), -2 = no room
XM_INL int condIntersect(const CondEnt* a, int na, const CondEnt* b, int nb, CondEnt* out, int cap) {
  int i = 0, j = 0, w = 0;
  while (i < na || j < nb) {
    CondEnt v;
    if (j >= nb || (i < na && (a[i] >> 2) < (b[j] >> 2))) v = a[i++];
    else if (i >= na || (b[j] >> 2) < (a[i] >> 2)) v = b[j++];
    else { if (a[i] != b[j]) return -1; v = a[i]; i++; j++; }
    if (w < cap) out[w] = v;
    w++;
  }
  return w <= cap ? w : -2;  // (a conflict behind the point where the room ended is still reported as a conflict)
}
struct MFrame { int32_t j, opt, condOff, condLen, found; };  // one activation of expand(): block index, next possibility, its startingCondition in MultiStore::stack
struct MultiStore {
  Poss* pool; int32_t poolUsed, poolCap;        // possibilities of the stored multi blocks; the mergeOptions of the block being made sit behind poolUsed until they are kept or dropped
  CondEnt* conds; int32_t condUsed, condCap;    // their conditions, same discipline
  CondEnt* stack; int32_t stackCap;             // startingConditions of the active expand() calls, one behind the other
  MFrame* frames; int32_t framesCap;
};
// Capacities at a scratch scale.  The fixed part is what the block being made can ask for whatever the scale (64 + 1 options from expand() + one per left
// possibility, and their conditions); the part that follows the scale is what is kept.  Sized so that the two mates of a pair, both with ambiguous bases,
// fit a read's region from scale 4 on beside the rest of the seeding state (at scale 1 a single read does).
struct MultiCaps { int32_t pool, conds, stack, frames; };
XM_INL MultiCaps multiCaps(int scale) {
  MultiCaps c;
  c.pool = 160 * scale + 2 * XM_MAX_COMBINATIONS + 8;
  c.conds = 640 * scale + 1024;
  c.stack = 128 * scale + 256;
  c.frames = 8 * scale + 24;
  return c;
}

struct Pyramid {
  SeqView seq;
  MultiStore* ms;       // null: the sequence is plain ACGT (every block is a single HashBlock)
  PBlock* blocks;       // levels >= 1, concatenated
  int32_t* levelStart;  // levelStart[k] (k>=1) = first block of level k; levelStart[nBuilt+1] = used
  int32_t nBuilt, used, cap, maxLevels;
  int32_t* status;
  DevCounters* dc;

  XM_INL void init(const SeqView& s, PBlock* b, int cap_, int32_t* ls, int maxLevels_, int32_t* st) {
    dc = nullptr;
    ms = nullptr;
    seq = s; blocks = b; cap = cap_; levelStart = ls; maxLevels = maxLevels_; nBuilt = 0; used = 0; status = st;
    levelStart[1] = 0;
  }

  // ---- multi mode: uniform view of a level's entries (level 0 is computed from the bases)
  XM_INL PBlock entryAt(int level, int i) const {
    if (level > 0) return blocks[levelStart[level] + i];
    uint8_t code = seq.at(i);
    PBlock b = level0Block(code, i);
    if (bpIsAmbiguous(code)) { b.flags = F_MULTI; b.fwd = 0; b.rev = 0; }
    return b;
  }
  XM_INL int numPoss(int level, const PBlock& e) const {
    if (!(e.flags & F_MULTI)) return 1;
    if (level > 0) return e.rev;
    return bpPop(seq.at(e.start));
  }
  XM_INL void possAt(int level, const PBlock& e, int k, PossView& p) const {  // getPossibilities()[k]
    if (!(e.flags & F_MULTI)) { p.block = e; p.cond = &p.own; p.condLen = 0; p.hasBlock = 1; return; }  // (this, ALWAYS), M/HashBlock.java:352-356
    if (level > 0) {
      const Poss& s = ms->pool[e.fwd + k];
      p.block = s.block; p.cond = ms->conds + s.condOff; p.condLen = s.condLen; p.hasBlock = s.hasBlock;
      return;
    }
    uint8_t code = seq.at(e.start);  // HashBlock_BaseRow.get: one possibility per base the code can stand for, in A C G T order
    int seen = 0;
    uint8_t option = 1;
    for (int bit = 0; bit < 4; bit++) {
      uint8_t o = (uint8_t)(1 << bit);
      if (code & o) { if (seen == k) { option = o; break; } seen++; }
    }
    p.block = level0Block(option, e.start);
    p.own = ((CondEnt)e.start << 2) | (CondEnt)(option == 1 ? 0 : option == 2 ? 1 : option == 4 ? 2 : 3);
    p.cond = &p.own; p.condLen = 1;
    p.hasBlock = 1;
  }
  // one more mergeOption behind the pool's kept possibilities; false = no room
  XM_INL bool multiAddOption(int& nOpt, int& nOptCond, const PBlock& block, bool hasBlock, const CondEnt* cond, int condLen) {
    if (ms->poolUsed + nOpt >= ms->poolCap || ms->condUsed + nOptCond + condLen > ms->condCap) return false;
    Poss& o = ms->pool[ms->poolUsed + nOpt];
    o.block = block; o.hasBlock = hasBlock ? 1 : 0; o.pad = 0;
    o.condOff = ms->condUsed + nOptCond; o.condLen = condLen;
    for (int t = 0; t < condLen; t++) ms->conds[o.condOff + t] = cond[t];
    nOpt++; nOptCond += condLen;
    return true;
  }
  // HashBlock_ParentRow.expand :137-191 for one left possibility; the recursion (a right possibility without a block passes the search
  // on to the block after it) runs on an explicit stack in the read's arena.  false = a capacity exceeded
  XM_INL bool multiExpand(int prev, int n, const PBlock& leftBlock, const CondEnt* leftCond, int leftCondLen, int i, int& nOpt, int& nOptCond) {
    MFrame* const st = ms->frames;
    CondEnt* const sc = ms->stack;
    if (leftCondLen > ms->stackCap || ms->framesCap < 1) return false;
    for (int t = 0; t < leftCondLen; t++) sc[t] = leftCond[t];
    int sp = 0;
    st[0].j = i; st[0].opt = 0; st[0].condOff = 0; st[0].condLen = leftCondLen; st[0].found = 0;
    while (sp >= 0) {
      MFrame& f = st[sp];
      const int nextIdx = f.j + 1;
      if (nextIdx >= n) { sp--; continue; }
      const PBlock next = entryAt(prev, nextIdx);
      if (f.opt >= numPoss(prev, next)) { sp--; continue; }
      PossView ro;
      possAt(prev, next, f.opt, ro);
      f.opt++;
      const int top = f.condOff + f.condLen;  // the intersection is written behind this call's condition: a nested call keeps it there
      const int icLen = condIntersect(sc + f.condOff, f.condLen, ro.cond, ro.condLen, sc + top, ms->stackCap - top);
      if (icLen == -2) return false;
      if (icLen < 0) { if (f.found) sp--; continue; }
      f.found = 1;
      if (nOpt > XM_MAX_COMBINATIONS) return true;  // every enclosing call returns at its next intersection without adding anything
      if (!ro.hasBlock) {
        if (sp + 1 >= ms->framesCap) return false;
        sp++;
        st[sp].j = nextIdx; st[sp].opt = 0; st[sp].condOff = top; st[sp].condLen = icLen; st[sp].found = 0;
        continue;
      }
      const bool merge = shouldMergeBlocks(leftBlock, ro.block);
      if (!multiAddOption(nOpt, nOptCond, merge ? mergeBlocks(leftBlock, ro.block) : leftBlock, merge, sc + top, icLen)) return false;
    }
    return true;
  }
  // one level in multi mode: maybeMakeBlock :69-127 for every block of the level below, in order
  XM_INL void buildLevelMulti() {
    const int prev = nBuilt;
    const int n = count(prev);
    int w = used;
    for (int i = 0; i + 1 < n; i++) {
      const PBlock L = entryAt(prev, i), R = entryAt(prev, i + 1);
      if (!((L.flags | R.flags) & F_MULTI)) {
        if (shouldMergeBlocks(L, R)) {
          if (w >= cap) { *status = XM_ST_OVERFLOW; return; }
          blocks[w++] = mergeBlocks(L, R);
        }
        continue;
      }
      int nOpt = 0, nOptCond = 0;
      const int nl = numPoss(prev, L);
      for (int k = 0; k < nl; k++) {
        PossView lo;
        possAt(prev, L, k, lo);
        if (lo.hasBlock) {
          if (!multiExpand(prev, n, lo.block, lo.cond, lo.condLen, i, nOpt, nOptCond)) { *status = XM_ST_OVERFLOW; return; }
        } else {
          if (!multiAddOption(nOpt, nOptCond, lo.block, false, lo.cond, lo.condLen)) { *status = XM_ST_OVERFLOW; return; }
        }
      }
      if (nOpt > 0 && nOpt <= XM_MAX_COMBINATIONS) {
        int minStart = -1, maxEnd = -1;
        for (int k = 0; k < nOpt; k++) if (ms->pool[ms->poolUsed + k].hasBlock) {
          const PBlock& b = ms->pool[ms->poolUsed + k].block;
          if (minStart < 0 || (int)b.start < minStart) minStart = b.start;
          if ((int)b.start + b.len > maxEnd) maxEnd = (int)b.start + b.len;
        }
        if (minStart >= 0) {  // hasNonEmpty: the options stay where they are
          if (w >= cap) { *status = XM_ST_OVERFLOW; return; }
          PBlock m;
          m.start = (uint16_t)minStart; m.len = (uint16_t)(maxEnd - minStart); m.fwd = ms->poolUsed; m.rev = nOpt; m.flags = F_MULTI; m.gapDir = 0; m.extraGap = 0;
          ms->poolUsed += nOpt; ms->condUsed += nOptCond;
          blocks[w++] = m;
        }
      }
    }
    nBuilt++;
    used = w;
    levelStart[nBuilt + 1] = w;
  }
  XM_INL int count(int level) const { return level == 0 ? seq.len : levelStart[level + 1] - levelStart[level]; }
  XM_INL PBlock blockAt(int level, int i) const { return level == 0 ? level0Block(seq.at(i), i) : blocks[levelStart[level] + i]; }

  XM_NOINL void ensureMulti(int level) {
    while (nBuilt < level) {
      if (nBuilt + 2 >= maxLevels) { *status = XM_ST_OVERFLOW; return; }
      buildLevelMulti();
      if (*status) return;
    }
  }
  XM_NOINL void ensure(int level) {  // M/HashBlock_Pyramid.java:15-24 + HashBlock_ParentRow.maybeMakeBlock
    if (ms) { ensureMulti(level); return; }  // (rare: kept out of the loop below)
    XM_TIC(t0);
    // locals: members reached through `this` would be re-loaded after every block store (possible aliasing)
    PBlock* const blk = blocks;
    const int capL = cap;
    const SeqView sq = seq;
    while (nBuilt < level) {
      if (nBuilt + 2 >= maxLevels) { *status = XM_ST_OVERFLOW; return; }
      int prev = nBuilt;
      int n = count(prev);
      int w = used;
      const int base = prev == 0 ? 0 : levelStart[prev];
      bool ovf = false;
      if (n > 1 && prev == 0) {  // level 1 from the bases themselves
        XM_GLOBAL(PBlockWord)* const dst = (XM_GLOBAL(PBlockWord)*)blk;
        PBlock L = level0Block(sq.at(0), 0);
        for (int i0 = 0; i0 + 1 < n && !ovf; i0 += 8) {
          uint8_t codes[8];
          const int m = imin(8, n - 1 - i0);
#pragma unroll
          for (int k = 0; k < 8; k++) codes[k] = sq.at(i0 + 1 + (k < m ? k : 0));
#pragma unroll
          for (int k = 0; k < 8; k++) {
            if (k < m && !ovf) {
              const PBlock R = level0Block(codes[k], i0 + 1 + k);
              if (shouldMergeBlocks(L, R)) {
                if (w >= capL) ovf = true;
                else dst[w++] = pbPack(mergeBlocks(L, R));
              }
              L = R;
            }
          }
        }
      } else if (n > 1) {
        // The new level is appended behind the level it is made from, so the blocks read never alias the blocks written: eight
        // reads are issued together (one memory round trip instead of eight dependent ones; the lane's arena is in HBM)
        XM_GLOBAL(const PBlockWord)* const src = (XM_GLOBAL(const PBlockWord)*)(blk + base);
        XM_GLOBAL(PBlockWord)* const dst = (XM_GLOBAL(PBlockWord)*)blk;
        PBlock L = pbUnpack(src[0]);
        for (int i0 = 0; i0 + 1 < n && !ovf; i0 += 8) {
          PBlockWord buf[8];
          const int m = imin(8, n - 1 - i0);
#pragma unroll
          for (int k = 0; k < 8; k++) buf[k] = src[i0 + 1 + (k < m ? k : 0)];
#pragma unroll
          for (int k = 0; k < 8; k++) {
            if (k < m && !ovf) {
              const PBlock R = pbUnpack(buf[k]);
              if (shouldMergeBlocks(L, R)) {
                if (w >= capL) ovf = true;
                else dst[w++] = pbPack(mergeBlocks(L, R));
              }
              L = R;
            }
          }
        }
      }
      if (ovf) { *status = XM_ST_OVERFLOW; return; }
      nBuilt++;
      used = w;
      levelStart[nBuilt + 1] = w;
    }
    XM_TOC(dc, T_PYRAMID, t0);
  }
  // first block of `level` whose start > pos (HashBlock_Row.getAfter)
  XM_INL bool getAfter(int level, int pos, PBlock& out) {
    if (level == 0) {
      int p = pos + 1;
      if (p >= seq.len) return false;
      out = ms ? entryAt(0, p) : level0Block(seq.at(p), p);
      return true;
    }
    ensure(level);
    if (*status) return false;
    int lo = levelStart[level], hi = levelStart[level + 1];
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if ((int)blocks[mid].start > pos) hi = mid; else lo = mid + 1;
    }
    if (lo >= levelStart[level + 1]) return false;
    out = blocks[lo];
    return true;
  }
  XM_INL bool get(int level, int index, PBlock& out) {  // HashBlock_Row.get
    if (level == 0) {
      if (index >= seq.len) return false;
      out = ms ? entryAt(0, index) : level0Block(seq.at(index), index);
      return true;
    }
    if (!getAfter(level, index - 1, out)) return false;
    return (int)out.start == index;
  }
};

// ---------------------------------------------------------------- database probes
XM_INL const Table* containingMap(const IndexView& ix, int used, int32_t* status) {
  if (used > ix.maxHashedLength) { *status = XM_ST_NEED_GROW; return nullptr; }  // M/Readable_HashBlock_Database.java:108-113
  return &ix.tables[used];
}
XM_INL uint32_t packedKey(const Table* t, int32_t key) {  // M/PackedMap.java:210-215
  int32_t r = key % t->capacity;
  if (r < 0) r += t->capacity;
  return (uint32_t)r;
}
// M/Readable_HashBlock_Database.java:72-80 + M/PackedMap.java:228-236: one 8-byte header probe (two adjacent CSR offsets)
XM_INL int numMatchesLowerBound(const IndexView& ix, const QBlock& b, DevCounters* dc, int32_t* status) {
  if (b.used < ix.minInterestingSize) return INT32_MAX;
  const Table* t = containingMap(ix, b.used, status);
  if (!t) return INT32_MAX;
  uint32_t k = packedKey(t, b.lookupKey());
  if (dc) dc->headerProbes++;
  if (ix.lines32 || ix.lines64) {  // bucket lines: the header is word 0 of the bucket's line
    const uint32_t h = ix.lines64 ? (uint32_t)ix.lines64[(t->offBase + k) * 8] : ix.lines32[(t->offBase + k) * 8];
    return (h & XM_OVERFULL) ? INT32_MAX : (int)h;
  }
  const uint32_t* off = ix.bucketOff + t->offBase + k;
  uint32_t o0 = off[0], o1 = off[1];
  if (o0 & XM_OVERFULL) return INT32_MAX;
  return (int)((o1 & ~XM_OVERFULL) - (o0 & ~XM_OVERFULL));
}
XM_INL int dbMaxNumMatchesAllowed(const IndexView& ix, const QBlock& b, int32_t* status) {  // :82-90
  if (b.used < ix.minInterestingSize) return -1;
  const Table* t = containingMap(ix, b.used, status);
  if (!t) return 0;
  return t->maxCount;
}
// a decoded SequencePosition
struct RefPos { int32_t contig; uint8_t rc; int32_t start; };
XM_INL RefPos decodePosition(const IndexView& ix, int64_t enc) {
  int lo = 0, hi = ix.numContigs * 2;  // last i with seqCumStart[i] <= enc
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (ix.seqCumStart[mid] <= enc) lo = mid; else hi = mid;
  }
  RefPos p;
  p.contig = lo >> 1;
  p.rc = (uint8_t)(lo & 1);
  p.start = (int32_t)(enc - ix.seqCumStart[lo]);
  return p;
}
// position `idx` of the virtual positions array: a word of a bucket line (XM_LINE_FLAG) or an entry of the CSR positions
XM_INL int64_t xmPositionAt(const IndexView& ix, int64_t idx) {
  if (idx & XM_LINE_FLAG) { idx &= ~XM_LINE_FLAG; return ix.lines64 ? (int64_t)ix.lines64[idx] : (int64_t)ix.lines32[idx]; }
  return ix.posIs64 ? (int64_t)ix.positions64[idx] : (int64_t)ix.positions32[idx];
}
// M/Readable_HashBlock_Database.java:22-38 / M/PackedMap.java:160-172.  returns -1 for Java null, else the hit count and
// (firstIndex into positions, invert flag)
XM_INL int matchBlock(const IndexView& ix, const QBlock& b, int64_t& first, bool& invert, DevCounters* dc, int32_t* status) {
  if (b.used < ix.minInterestingSize) return -1;
  const Table* t = containingMap(ix, b.used, status);
  if (!t) return -1;
  uint32_t k = packedKey(t, b.lookupKey());
  if (dc) { dc->headerProbes++; dc->bucketFetches++; }
  int count;
  if (ix.lines32 || ix.lines64) {
    // bucket lines: the access that returns the count has brought the first XM_LINE_SLOTS positions with it (fetchHit reads them from the line)
    const int64_t line = (t->offBase + k) * 8;
    const uint32_t h = ix.lines64 ? (uint32_t)ix.lines64[line] : ix.lines32[line];
    if (h & XM_OVERFULL) return -1;
    count = (int)h;
    if (count > t->maxCount) return -1;
    if (count <= XM_LINE_SLOTS) first = XM_LINE_FLAG | (line + 1);
    else first = t->posBase + (int64_t)(ix.bucketOff[t->offBase + k] & ~XM_OVERFULL);
  } else {
    const uint32_t* off = ix.bucketOff + t->offBase + k;
    uint32_t o0 = off[0], o1 = off[1];
    if (o0 & XM_OVERFULL) return -1;
    count = (int)((o1 & ~XM_OVERFULL) - (o0 & ~XM_OVERFULL));
    if (count > t->maxCount) return -1;
    first = t->posBase + (int64_t)(o0 & ~XM_OVERFULL);
  }
  invert = !b.isPrimaryPolarity();
  if (dc) dc->hitsFetched += (unsigned long long)count;
  return count;
}
XM_INL RefPos fetchHit(const IndexView& ix, int64_t idx, bool invert, int blockSpan) {
  int64_t enc = xmPositionAt(ix, idx);
  RefPos p = decodePosition(ix, enc);
  if (invert) {  // Readable_HashBlock_Database.reverseComplement :55-59
    p.start = ix.contigLen[p.contig] - p.start - blockSpan;
    p.rc ^= 1;
  }
  return p;
}

// ---------------------------------------------------------------- HashBlockPath (M/HashBlockPath.java)
struct PathState {
  int32_t batchIndex;
  bool curExists;
  PBlock cur;
  bool gapComputed;
  int32_t gapStatus;
  QBlock gap;
  bool havePrev1, havePrev2;
  int32_t prevFwd1, prevFwd2;
};

// ---------------------------------------------------------------- voting
struct Counter {  // M/HashBlockMatch_Counter.java + its SequenceMatch
  int32_t offset;
  int32_t contig;
  int32_t numMatches, numDistinctMismatches, lastMismatchedPosition, lastMatchedBlockId, historyProcessedIndex, priority;
  int32_t next, prev;  // neighbouring counters within maxIndelLengthToConsider (index or -1)
  uint8_t seqAId;      // identity of sequenceA (the path's query or its reverse complement)
  uint8_t mapSel;      // 0: forwardMatchCounters (match.getReversed()), 1: reverseMatchCounters   (sic, M/Counting_HashBlockPath.java:197-200)
  uint8_t good;
};

struct SeqMatch {  // M/SequenceMatch.java
  int32_t offset, contig;
  uint8_t seqAId;
  XM_INL bool reversed() const { return (seqAId & 1) != 0; }
};

struct QMatch {  // M/QueryMatch.java
  int32_t n, priority;
  SeqMatch c[2];
  uint8_t hint;
};

struct ListRef { int32_t id, n; int16_t* items; };

struct ReadCtx;  // xm_worker.h

struct Comp {  // one Counting_HashBlockPath (+ its HashBlockPath and pyramid)
  Pyramid pyr;
  PathState path;
  SeqView query, rcQuery;
  Counter* counters; int32_t nCounters;
  int16_t* good; int32_t nGood;
  bool foundGood, done;
  QBlock* history; int32_t nHistory;
  QBlock* pending; int32_t pendHead, pendTail;
  int32_t numBlocksMatchingAnywhere, maxNonoverlappingBlockVisited, numNonoverlappingBlocksVisited, minNumDistinctMismatches;
  int32_t maxIndelLengthToConsider;
  int32_t nextBlockId;
  ListRef hp, best, all;  // previousHighPriorityMatchCounters / getBestMatches() / previousAllPositions
};

struct SeedEnv {  // what the seed code needs from the surrounding read context
  const IndexView* ix;
  const Caps* caps;
  DevCounters* dc;
  int32_t* status;
  int32_t* listIdCounter;
  const int32_t* mateLen;  // [2]
};

XM_INL int seqALen(const SeedEnv& e, uint8_t seqAId) { return e.mateLen[seqAId >> 1]; }
XM_INL int smStartB(const SeqMatch& m) { return imax(0, m.offset); }
XM_INL int smEndB(const SeedEnv& e, const SeqMatch& m) { return imin(m.offset + seqALen(e, m.seqAId), e.ix->contigLen[m.contig]); }
XM_INL bool smEquals(const SeqMatch& a, const SeqMatch& b) { return a.offset == b.offset && a.seqAId == b.seqAId && a.contig == b.contig; }

XM_INL void pathInit(PathState& p) {  // M/HashBlockPath.java:15-24
  p.batchIndex = -1;
  p.curExists = true;
  p.cur.start = 0; p.cur.len = 0; p.cur.fwd = 0; p.cur.rev = 0; p.cur.flags = 0; p.cur.gapDir = 0; p.cur.extraGap = 0;
  p.gapComputed = false;
  p.gapStatus = 0;
  p.havePrev1 = p.havePrev2 = false;
  p.prevFwd1 = p.prevFwd2 = 0;
}

XM_INL void pathMoveRight(Comp& c) {  // :125-128
  PBlock nb;
  c.path.curExists = c.pyr.getAfter(c.path.batchIndex, c.path.cur.start, nb);
  if (c.path.curExists) c.path.cur = nb;
  c.path.gapComputed = false;
}
XM_INL void pathMoveDown(Comp& c) {  // :99-108
  c.path.batchIndex--;
  PBlock nb;
  c.path.curExists = c.pyr.getAfter(c.path.batchIndex, c.path.cur.start, nb);
  if (c.path.curExists) c.path.cur = nb;
  c.path.gapComputed = false;
}
XM_INL void pathMoveUpOrRight(Comp& c) {  // :111-122
  PBlock up;
  if (c.pyr.get(c.path.batchIndex + 1, c.path.cur.start, up)) {
    c.path.batchIndex++;
    c.path.cur = up;
    c.path.gapComputed = false;
  } else {
    pathMoveRight(c);
  }
}
// :197-203.  returns false for null
XM_INL bool pathWithGap(Comp& c, const SeedEnv& e, QBlock& out) {
  if (!e.ix->enableGapmers) {
    const PBlock& b = c.path.cur;
    out.start = b.start; out.len = b.len; out.used = b.len; out.fwd = b.fwd; out.rev = b.rev; out.flags = b.flags; out.id = -1;
    return true;
  }
  if (!c.path.gapComputed) {
    c.path.gapStatus = withGapAndExtension(c.path.cur, c.pyr.seq, c.path.gap);
    c.path.gap.id = -1;
    c.path.gapComputed = true;
  }
  if (c.path.gapStatus == 0) return false;
  out = c.path.gap;
  return true;
}
XM_INL int pathMaxNumMatchesAllowed(Comp& c, const SeedEnv& e, const QBlock& b) {  // :205-219
  if (b.len >= c.query.len / 6) return dbMaxNumMatchesAllowed(*e.ix, b, e.status);
  if (b.flags & F_RMR) return 5;
  return b.used + 1;
}
// :143-195.  returns false when the path is exhausted
XM_NOINL_W1 bool pathAdvance(Comp& c, const SeedEnv& e) {
  XM_TIC(t0);
  int singleLen = c.path.cur.len;
  if (maxGapmerNumBasepairsUsed(singleLen) < e.ix->minInterestingSize && e.ix->enableGapmers) {
    pathMoveUpOrRight(c);
  } else {
    QBlock ext;
    if (pathWithGap(c, e, ext)) {
      int numMatches = numMatchesLowerBound(*e.ix, ext, e.dc, e.status);
      if (numMatches < 6) {
        if (c.path.batchIndex > 0) pathMoveDown(c); else pathMoveRight(c);
      } else {
        if (numMatches > pathMaxNumMatchesAllowed(c, e, ext)) pathMoveUpOrRight(c);
        else pathMoveRight(c);
      }
    } else {
      int typical = singleLen * 3 / 2;
      if (typical <= e.ix->minInterestingSize && e.ix->enableGapmers) pathMoveUpOrRight(c);
      else { if (c.path.batchIndex > 0) pathMoveDown(c); else pathMoveRight(c); }
    }
  }
  // skipMultiblocks :130-140 (only a read with ambiguous bases has such blocks)
  while (c.path.curExists && (c.path.cur.flags & F_MULTI) && *e.status == 0) {
    if (c.path.batchIndex > 0) pathMoveDown(c); else pathMoveRight(c);
  }
  XM_TOC(e.dc, T_WALK, t0);
  return c.path.curExists && *e.status == 0;
}
// getNextInterestingBlock :27-50 (+ getNextBlockWithGoodNumberOfMatches :68-96, recentlySeen :52-65)
XM_NOINL_W1 bool pathNextInterestingBlock(Comp& c, const SeedEnv& e, QBlock& out) {
  if (!c.path.curExists) return false;
  while (true) {
    if (!pathAdvance(c, e)) return false;
    QBlock ext;
    if (!pathWithGap(c, e, ext)) continue;
    int n = numMatchesLowerBound(*e.ix, ext, e.dc, e.status);
    if (*e.status) return false;
    if (!(n <= pathMaxNumMatchesAllowed(c, e, ext))) continue;
    bool seen = false;
    if (c.path.havePrev1 && ext.fwd == c.path.prevFwd1) seen = true;
    else if (c.path.havePrev2 && ext.fwd == c.path.prevFwd2) seen = true;
    c.path.havePrev2 = c.path.havePrev1;
    c.path.prevFwd2 = c.path.prevFwd1;
    c.path.havePrev1 = true;
    c.path.prevFwd1 = ext.fwd;
    if (seen) continue;
    ext.id = c.nextBlockId++;
    out = ext;
    return true;
  }
}

// ---------------------------------------------------------------- Counting_HashBlockPath
XM_INL void counterUpdate(Comp& c, const SeedEnv& e, Counter& k) {  // M/HashBlockMatch_Counter.java:41-46,74-88
  while (k.historyProcessedIndex < c.nHistory) {
    const QBlock& b = c.history[k.historyProcessedIndex];
    if (b.id != k.lastMatchedBlockId) {
      if (b.start >= k.lastMismatchedPosition) {
        if (k.offset + b.end() <= e.ix->contigLen[k.contig]) {
          k.numDistinctMismatches++;
          k.lastMismatchedPosition = b.end();
        }
      }
    }
    k.historyProcessedIndex++;
  }
}
XM_INL int counterNumDistinctMismatches(Comp& c, const SeedEnv& e, Counter& k) { counterUpdate(c, e, k); return k.numDistinctMismatches; }
XM_INL void declareGood(Comp& c, const SeedEnv& e, int ci) {  // M/Counting_HashBlockPath.java:280-285
  Counter& k = c.counters[ci];
  if (!k.good) {
    if (c.nGood >= e.caps->maxCounters) { *e.status = XM_ST_OVERFLOW; return; }
    c.good[c.nGood++] = (int16_t)ci;
    k.good = 1;
    k.priority = counterNumDistinctMismatches(c, e, k);  // setGood
  }
}
XM_INL void compAddMatch(Comp& c, const SeedEnv& e, int ci, const QBlock& qb, int queryBlockNumMatches, const SeqMatch& fullMatch) {  // :254-277
  Counter& k = c.counters[ci];
  k.numMatches++;
  k.lastMatchedBlockId = qb.id;
  counterUpdate(c, e, k);
  if (k.numMatches <= 1) {
    if (k.numMatches == 1) {
      c.foundGood = true;
      declareGood(c, e, ci);
    } else {
      if (queryBlockNumMatches <= qb.len) {
        int distanceFromStart = fullMatch.offset;
        int distanceFromEnd = e.ix->contigLen[fullMatch.contig] - (fullMatch.offset + seqALen(e, fullMatch.seqAId));
        if (imin(distanceFromStart, distanceFromEnd) < 0) declareGood(c, e, ci);
      }
    }
  }
}
XM_NOINL_W3 void compUpdateMatches(Comp& c, const SeedEnv& e, const SeqMatch& m, const QBlock& qb, int queryBlockNumMatches) {  // :193-252
  uint8_t mapSel = m.reversed() ? 0 : 1;
  int cur = -1, lower = -1, higher = -1;
  for (int i = 0; i < c.nCounters; i++) {
    const Counter& k = c.counters[i];
    if (k.mapSel != mapSel || k.contig != m.contig) continue;
    if (k.offset == m.offset) { cur = i; break; }
    if (k.offset < m.offset) { if (lower < 0 || k.offset > c.counters[lower].offset) lower = i; }
    else { if (higher < 0 || k.offset < c.counters[higher].offset) higher = i; }
  }
  if (cur < 0) {
    if (c.nCounters >= e.caps->maxCounters) { *e.status = XM_ST_OVERFLOW; return; }
    cur = c.nCounters++;
    Counter& k = c.counters[cur];
    k.offset = m.offset; k.contig = m.contig; k.seqAId = m.seqAId; k.mapSel = mapSel; k.good = 0;
    k.numMatches = 0;
    k.numDistinctMismatches = c.numNonoverlappingBlocksVisited;
    k.lastMismatchedPosition = qb.start;
    k.lastMatchedBlockId = -2;
    k.historyProcessedIndex = c.nHistory - 1;
    k.priority = 0;
    k.next = k.prev = -1;
    if (lower >= 0 && iabs(c.counters[lower].offset - m.offset) <= c.maxIndelLengthToConsider) {
      k.prev = lower;
      c.counters[lower].next = cur;
    }
    if (higher >= 0 && iabs(c.counters[higher].offset - m.offset) <= c.maxIndelLengthToConsider) {
      k.next = higher;
      c.counters[higher].prev = cur;
    }
  }
  int prev = c.counters[cur].prev;
  if (prev >= 0) compAddMatch(c, e, prev, qb, queryBlockNumMatches, m);
  int next = c.counters[cur].next;
  if (next >= 0) compAddMatch(c, e, next, qb, queryBlockNumMatches, m);
  bool updateThisOne = true;
  if ((prev >= 0 && c.counters[prev].good) || (next >= 0 && c.counters[next].good)) {
    if (!c.counters[cur].good) updateThisOne = false;
  }
  if (updateThisOne) compAddMatch(c, e, cur, qb, queryBlockNumMatches, m);
}

// iterate counters of one map in (contig, offset) order: returns the next index after (lastContig,lastOffset) or -1
XM_INL int nextCounterInOrder(const Comp& c, uint8_t mapSel, int lastContig, int lastOffset, bool first) {
  int best = -1;
  for (int i = 0; i < c.nCounters; i++) {
    const Counter& k = c.counters[i];
    if (k.mapSel != mapSel) continue;
    if (!first && (k.contig < lastContig || (k.contig == lastContig && k.offset <= lastOffset))) continue;
    if (best < 0 || k.contig < c.counters[best].contig || (k.contig == c.counters[best].contig && k.offset < c.counters[best].offset)) best = i;
  }
  return best;
}
XM_NOINL_W2 void tryEnsureGoodMatchCounter(Comp& c, const SeedEnv& e) {  // :291-308
  if (!c.foundGood && c.nCounters <= c.query.len) {
    for (int mapSel = 0; mapSel < 2; mapSel++) {
      int lc = 0, lo = 0;
      bool first = true;
      while (true) {
        int i = nextCounterInOrder(c, (uint8_t)mapSel, lc, lo, first);
        if (i < 0) break;
        first = false; lc = c.counters[i].contig; lo = c.counters[i].offset;
        declareGood(c, e, i);
      }
    }
    c.foundGood = true;
  }
}

// getNextInterestingBlock :344-368
XM_NOINL_W2 bool compNextInterestingBlock(Comp& c, const SeedEnv& e, QBlock& out) {
  c.all.id = 0;  // previousAllPositions = null
  while (true) {
    QBlock b;
    if (!pathNextInterestingBlock(c, e, b)) {
      if (*e.status) return false;
      if (c.pendHead >= c.pendTail) return false;
      out = c.pending[c.pendHead++];
      return true;
    }
    if (b.start < c.maxNonoverlappingBlockVisited) {
      if (c.pendTail >= e.caps->maxPending) { *e.status = XM_ST_OVERFLOW; return false; }
      c.pending[c.pendTail++] = b;
      continue;
    }
    out = b;
    return true;
  }
}

// step() :40-179
XM_NOINL_W4 bool compStep(Comp& c, const SeedEnv& e) {
  if (c.done) return false;
  QBlock qb;
  int64_t first = 0;
  bool invert = false;
  int nHits;
  while (true) {  // getNextInterestingMatch :371-384
    if (!compNextInterestingBlock(c, e, qb)) {
      if (*e.status) return false;
      c.done = true;
      if (c.numBlocksMatchingAnywhere < 1) tryEnsureGoodMatchCounter(c, e);
      return false;
    }
    nHits = matchBlock(*e.ix, qb, first, invert, e.dc, e.status);
    if (*e.status) return false;
    if (nHits < 0) continue;
    break;
  }
  if (c.nHistory >= e.caps->maxHistory) { *e.status = XM_ST_OVERFLOW; return false; }
  c.history[c.nHistory++] = qb;
  const IndexView& ix = *e.ix;
  XM_TIC(tHits);
  for (int h = 0; h < nHits; h++) {
    RefPos rp = fetchHit(ix, first + h, invert, qb.len);
    SeqView refSeq = refView(ix, rp.contig, rp.rc != 0);
    int numMismatchedItems = 0, numMatchedItems = 0;
    for (int distance = 1; distance < 20; distance++) {  // :98-148 flank vote
      int checkOffset = -distance;
      int queryIndex = qb.start + checkOffset;
      if (queryIndex >= 0 && queryIndex < c.query.len) {
        int referenceIndex = rp.start + checkOffset;
        if (referenceIndex >= 0 && referenceIndex < refSeq.len) {
          if (!bpCanMatch(c.query.at(queryIndex), refSeq.at(referenceIndex))) numMismatchedItems++; else numMatchedItems++;
        }
      }
      checkOffset = qb.len - 1 + distance;
      queryIndex = qb.start + checkOffset;
      if (queryIndex >= 0 && queryIndex < c.query.len) {
        int referenceIndex = rp.start + checkOffset;
        if (referenceIndex >= 0 && referenceIndex < refSeq.len) {
          if (!bpCanMatch(c.query.at(queryIndex), refSeq.at(referenceIndex))) numMismatchedItems++; else numMatchedItems++;
        }
      }
      if (numMatchedItems < numMismatchedItems) break;
      if (numMatchedItems >= numMismatchedItems + qb.used) break;
    }
    if (numMismatchedItems > numMatchedItems) continue;
    SeqMatch fm;
    fm.contig = rp.contig;
    if (rp.rc) {  // :155-161
      int reverseQueryBlockStart = c.query.len - qb.end();
      int reverseReferenceBlockStart = refSeq.len - (rp.start + qb.len);
      fm.offset = reverseReferenceBlockStart - reverseQueryBlockStart;
      fm.seqAId = c.rcQuery.id;
    } else {
      fm.offset = rp.start - qb.start;
      fm.seqAId = c.query.id;
    }
    compUpdateMatches(c, e, fm, qb, nHits);
    if (*e.status) return false;
  }
  XM_TOC(e.dc, T_HITS, tHits);
  if (qb.start >= c.maxNonoverlappingBlockVisited) {
    c.maxNonoverlappingBlockVisited = qb.end();
    c.numNonoverlappingBlocksVisited++;
  }
  c.numBlocksMatchingAnywhere++;
  c.minNumDistinctMismatches = -1;
  return true;
}

XM_NOINL ListRef compFindGoodPositionsHavingPriorityUpTo(Comp& c, const SeedEnv& e, int priority) {  // :406-433
  while (true) {
    if (c.numNonoverlappingBlocksVisited >= jadd(priority, 1)) break;
    if (!compStep(c, e)) break;
  }
  if (c.hp.id != 0 && c.hp.n == c.nGood) return c.hp;
  int n = 0;
  for (int i = 0; i < c.nGood; i++) {
    if (c.counters[c.good[i]].priority <= priority) c.hp.items[n++] = c.good[i];
  }
  c.hp.n = n;
  c.hp.id = ++(*e.listIdCounter);
  return c.hp;
}
XM_NOINL ListRef compGetAllPositions(Comp& c, const SeedEnv& e) {  // :435-451
  if (c.all.id == 0) {
    int n = 0;
    for (int mapSel = 0; mapSel < 2; mapSel++) {
      int lc = 0, lo = 0;
      bool first = true;
      while (true) {
        int i = nextCounterInOrder(c, (uint8_t)mapSel, lc, lo, first);
        if (i < 0) break;
        first = false; lc = c.counters[i].contig; lo = c.counters[i].offset;
        c.all.items[n++] = (int16_t)i;
      }
    }
    c.all.n = n;
    c.all.id = ++(*e.listIdCounter);
  }
  return c.all;
}
XM_NOINL ListRef compGetBestMatches(Comp& c, const SeedEnv& e) {  // :471-493 (+ getNumGoodDistinctMismatches :457-469)
  c.best.id = ++(*e.listIdCounter);
  c.best.n = 0;
  if (c.numBlocksMatchingAnywhere < 1) return c.best;
  if (c.minNumDistinctMismatches < 0) {
    int mn = c.numNonoverlappingBlocksVisited - 1;
    for (int i = 0; i < c.nGood; i++) {
      int cnt = counterNumDistinctMismatches(c, e, c.counters[c.good[i]]);
      if (mn >= cnt) mn = cnt;
    }
    c.minNumDistinctMismatches = mn;
  }
  int mn = c.minNumDistinctMismatches;
  int n = 0;
  for (int i = 0; i < c.nGood; i++) {
    int cnt = counterNumDistinctMismatches(c, e, c.counters[c.good[i]]);
    if (cnt <= mn) c.best.items[n++] = c.good[i];
  }
  c.best.n = n;
  return c.best;
}

// ---------------------------------------------------------------- HashBlockPaths_Counter (M/HashBlockPaths_Counter.java)
struct PathsCounter {
  Comp* comps;
  int32_t nComps;
  int32_t maxOffsetBetweenComponents;
  bool foundNonemptyResult;
  bool havePrevious;
  int32_t prevListId[2];
  QMatch* assembled; int32_t nAssembled;  // previousAssembledMatches
  QMatch* filtered; int32_t nFiltered;    // result of the last filter* call
  int16_t* nearby;                         // scratch [maxCounters]
};

XM_INL SeqMatch counterMatch(const Counter& k) { SeqMatch m; m.offset = k.offset; m.contig = k.contig; m.seqAId = k.seqAId; return m; }

XM_INL int countPriority(const PathsCounter& pc, const SeedEnv& e, int c0, int c1) {  // :314-334 (2 counters)
  const Counter& a = pc.comps[0].counters[c0];
  const Counter& b = pc.comps[1].counters[c1];
  SeqMatch m1 = counterMatch(a), m2 = counterMatch(b);
  if (smStartB(m1) < smEndB(e, m2) && smEndB(e, m1) > smStartB(m2)) return imax(imax(0, a.priority), b.priority);
  return a.priority + b.priority;
}

XM_NOINL void pcMatchWithoutCache(PathsCounter& pc, const SeedEnv& e, const ListRef* lists) {  // :136-247 + assembleQueryMatches :249-265
  pc.nAssembled = 0;
  if (pc.nComps == 1) {
    for (int i = 0; i < lists[0].n; i++) {
      if (pc.nAssembled >= e.caps->maxQM) { *e.status = XM_ST_OVERFLOW; return; }
      const Counter& k = pc.comps[0].counters[lists[0].items[i]];
      QMatch& q = pc.assembled[pc.nAssembled++];
      q.n = 1; q.priority = k.priority; q.c[0] = counterMatch(k); q.hint = 0;
    }
    return;
  }
  bool lastComponentIsLargest = lists[0].n <= lists[1].n;
  int firstComp = lastComponentIsLargest ? 0 : 1;
  int secondComp = 1 - firstComp;
  Comp& FC = pc.comps[firstComp];
  Comp& SC = pc.comps[secondComp];
  for (int j = 0; j < lists[secondComp].n; j++) {
    int ci = lists[secondComp].items[j];
    const Counter& k = SC.counters[ci];
    int querySequenceLength = seqALen(e, k.seqAId);
    int maxReverseOffset = querySequenceLength / 2;
    bool sequenceMatchReversed = (k.seqAId & 1) != 0;
    bool queryMatchReversed = (sequenceMatchReversed == (secondComp % 2 == 0));
    int offset = k.offset;
    int searchStart, searchEnd;
    bool otherSequenceExpectEarlier = (queryMatchReversed == lastComponentIsLargest);
    if (otherSequenceExpectEarlier) { searchStart = offset - maxReverseOffset; searchEnd = jadd(offset, pc.maxOffsetBetweenComponents); }
    else { searchStart = offset - pc.maxOffsetBetweenComponents; searchEnd = offset + maxReverseOffset; }
    if (searchStart > searchEnd) { *e.status = XM_ST_INTERNAL; return; }  // TreeMap.subMap would throw
    // entries of the first component filed under the same (direction, contig), offset in [searchStart, searchEnd], ascending
    int nn = 0;
    for (int i = 0; i < lists[firstComp].n; i++) {
      int fi = lists[firstComp].items[i];
      const Counter& f = FC.counters[fi];
      bool fRev = (f.seqAId & 1) != 0;
      bool fQueryMatchReversed = (fRev == (firstComp % 2 == 0));
      if (fQueryMatchReversed != queryMatchReversed || f.contig != k.contig) continue;
      if (f.offset < searchStart || f.offset > searchEnd) continue;
      int p = nn++;
      while (p > 0 && FC.counters[pc.nearby[p - 1]].offset > f.offset) { pc.nearby[p] = pc.nearby[p - 1]; p--; }
      pc.nearby[p] = (int16_t)fi;
    }
    bool descending = queryMatchReversed && nn > 1;
    for (int t = 0; t < nn; t++) {
      int fi = pc.nearby[descending ? nn - 1 - t : t];
      int c0 = lastComponentIsLargest ? fi : ci;  // counter of component 0
      int c1 = lastComponentIsLargest ? ci : fi;  // counter of component 1
      if (pc.nAssembled >= e.caps->maxQM) { *e.status = XM_ST_OVERFLOW; return; }
      QMatch& q = pc.assembled[pc.nAssembled++];
      q.n = 2;
      q.c[0] = counterMatch(pc.comps[0].counters[c0]);
      q.c[1] = counterMatch(pc.comps[1].counters[c1]);
      q.hint = counterNumDistinctMismatches(pc.comps[0], e, pc.comps[0].counters[c0]) < counterNumDistinctMismatches(pc.comps[1], e, pc.comps[1].counters[c1]) ? 1 : 0;
      q.priority = countPriority(pc, e, c0, c1);
    }
  }
}
XM_INL void pcMatch(PathsCounter& pc, const SeedEnv& e, const ListRef* lists) {  // :116-133
  bool same = pc.havePrevious;
  if (same) for (int i = 0; i < pc.nComps; i++) if (pc.prevListId[i] != lists[i].id) { same = false; break; }
  if (!same) {
    pcMatchWithoutCache(pc, e, lists);
    for (int i = 0; i < pc.nComps; i++) pc.prevListId[i] = lists[i].id;
    pc.havePrevious = true;
  }
}
XM_INL void pcFilterPriority(PathsCounter& pc, const QMatch* src, int n, int priority) {  // :267-294
  pc.nFiltered = 0;
  for (int i = 0; i < n; i++) if (src[i].priority == priority) pc.filtered[pc.nFiltered++] = src[i];
}
XM_NOINL void pcFindGoodPositionsHavingPriority(PathsCounter& pc, const SeedEnv& e, int numMismatches) {  // :21-24, :51-81
  ListRef lists[2];
  for (int i = 0; i < pc.nComps; i++) {
    lists[i] = compFindGoodPositionsHavingPriorityUpTo(pc.comps[i], e, numMismatches);
    if (*e.status) { pc.nFiltered = 0; return; }
    if (lists[i].n >= 1) pc.foundNonemptyResult = true;
  }
  pcMatch(pc, e, lists);
  if (*e.status) { pc.nFiltered = 0; return; }
  pcFilterPriority(pc, pc.assembled, pc.nAssembled, numMismatches);
}
XM_NOINL void pcOptimisticGetBestMatches(PathsCounter& pc, const SeedEnv& e) {  // :84-98 (+ filterMatchesHavingMinPriority :296-304, sic: max)
  ListRef lists[2];
  for (int i = 0; i < pc.nComps; i++) {
    while (true) {
      ListRef best = compGetBestMatches(pc.comps[i], e);
      if (best.n == 1 || !compStep(pc.comps[i], e)) { lists[i] = best; break; }
    }
    if (*e.status) { pc.nFiltered = 0; return; }
  }
  pcMatch(pc, e, lists);
  if (*e.status) { pc.nFiltered = 0; return; }
  int mn = -1;
  for (int i = 0; i < pc.nAssembled; i++) if (mn < 0 || mn < pc.assembled[i].priority) mn = pc.assembled[i].priority;
  pcFilterPriority(pc, pc.assembled, pc.nAssembled, mn);
}
XM_NOINL void pcFindPartiallyGoodPositions(PathsCounter& pc, const SeedEnv& e) {  // :26-49
  pc.nFiltered = 0;
  if (pc.nComps != 2) return;
  if (!pc.foundNonemptyResult) return;
  ListRef lists[2];
  bool foundGoodPosition = false, foundBadPosition = false;
  for (int i = 0; i < 2; i++) {
    ListRef here = compFindGoodPositionsHavingPriorityUpTo(pc.comps[i], e, INT32_MAX);
    if (*e.status) return;
    if (here.n == 0) { foundBadPosition = true; here = compGetAllPositions(pc.comps[i], e); }
    else foundGoodPosition = true;
    lists[i] = here;
  }
  if (foundGoodPosition && foundBadPosition) {
    pcMatch(pc, e, lists);
    if (*e.status) return;
    for (int i = 0; i < pc.nAssembled; i++) pc.filtered[pc.nFiltered++] = pc.assembled[i];
  }
}
XM_INL int pcGetNumBlocks(const PathsCounter& pc) {  // :108-114
  int t = 0;
  for (int i = 0; i < pc.nComps; i++) t += pc.comps[i].numBlocksMatchingAnywhere;
  return t;
}

// QueryMatch helpers (M/QueryMatch.java)
XM_INL bool qmReversed(const QMatch& q) { return q.c[0].reversed(); }
XM_INL int qmQueryTotalLength(const SeedEnv& e, const QMatch& q) { int t = 0; for (int i = 0; i < q.n; i++) t += seqALen(e, q.c[i].seqAId); return t; }
XM_INL int qmStartIndexB(const QMatch& q) { return imin(smStartB(q.c[0]), smStartB(q.c[q.n - 1])); }
XM_INL int qmEndIndexB(const QMatch& q) { return imax(smStartB(q.c[0]), smStartB(q.c[q.n - 1])); }  // (sic) :54-58
XM_INL int qmTotalDistanceBetweenComponents(const SeedEnv& e, const QMatch& q) {  // :70-79,123-132
  int total = 0;
  for (int i = 1; i < q.n; i++) {
    const SeqMatch& a = q.c[i - 1];
    const SeqMatch& b = q.c[i];
    int d;
    if (a.contig != b.contig) d = INT32_MAX;
    else if (qmReversed(q)) d = smStartB(a) - smEndB(e, b);
    else d = smStartB(b) - smEndB(e, a);
    total = jadd(total, d);
  }
  return total;
}
XM_INL bool qmSamePosition(const QMatch& a, const QMatch& b) {  // :81-93
  if (a.n != b.n) return false;
  for (int i = 0; i < a.n; i++) if (!smEquals(a.c[i], b.c[i])) return false;
  return true;
}

}  // namespace xm
