// libxm_hostio.so: host-side I/O of the standalone harness (`python -m mapper_amd`) in native code - FASTA / FASTQ (plain or .gz) into the flat
// batch arrays of include/xmapper_hip.h without one object per read, and the result streams of a batch into SAM text (and the unaligned-query file) without
// one object per alignment.  SURVEY.md section 8(f) rank 1: the harness around the accelerated path.  In the drop-in deployment the Java host keeps doing
// all of this (north_star: "all I/O stays Java"): Mapper.java:699-732 hands QueryAlignments to its SamWriter; this library is what lets the Python harness
// feed a kernel that aligns millions of reads per second.  Plain C++17 for the host, no GPU code, no part of the alignment path.
//
// What the reference pins of the formats: the five SAM bodies of SamWriter_Test.java:18-94 (flags 0 / 99 / 147 / 73, MAPQ 255, column 9 = read length, RNEXT =
// contig name, mate 2 printed as aligned, `cs:f:` only for paired queries, `AS:f:` = Double.toString of the penalty), the section rule of
// SequenceSplitter.java:9-38, the statistics of Mapper.run :786-796.  What they do not show follows the SAM specification and is marked [unpinned] in
// mapper_amd/sam.py, whose records() this formatter reproduces byte for byte (tests/test_hostio.py compares the two on random result streams).
#include "../../include/xmapper_hostio.h"
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>
#include <fcntl.h>

namespace {

thread_local std::string g_error;
int fail(const std::string& m) { g_error = m; return -1; }

uint8_t g_code[256];
uint8_t g_comp[16];
const char g_decode[17] = "?ACMGRSVTWYHKDBN";
struct Tables {
  Tables() {
    memset(g_code, 15, sizeof(g_code));
    const char* letters = "ACGTURYSWKMBDHVN";
    const uint8_t values[16] = {1, 2, 4, 8, 8, 5, 10, 6, 9, 12, 3, 14, 13, 11, 7, 15};
    for (int i = 0; i < 16; i++) { g_code[(uint8_t)letters[i]] = values[i]; g_code[(uint8_t)(letters[i] | 0x20)] = values[i]; }
    for (int b = 0; b < 16; b++) g_comp[b] = (uint8_t)(((b & 1) << 3) | ((b & 2) << 1) | ((b & 4) >> 1) | ((b & 8) >> 3));
  }
} g_tables;

// ---------------------------------------------------------------- reading
struct Source {
  gzFile f = nullptr;
  std::string path;
  std::vector<char> buf;
  size_t pos = 0, end = 0;
  bool eof = false;
  long long line = 0;
  int fd = -1;  // an uncompressed file is read directly (zlib's transparent mode would copy every byte once more)
  bool open(const char* p) {
    path = p;
    buf.resize(8 << 20);
    FILE* probe = fopen(p, "rb");
    if (!probe) return false;
    unsigned char magic[2] = {0, 0};
    const size_t got = fread(magic, 1, 2, probe);
    fclose(probe);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
      f = gzopen(p, "rb");
      if (!f) return false;
      gzbuffer(f, 1 << 20);
    } else {
      fd = ::open(p, O_RDONLY);
      if (fd < 0) return false;
    }
    return true;
  }
  void close() { if (f) gzclose(f); f = nullptr; if (fd >= 0) ::close(fd); fd = -1; }
  bool fill() {  // keeps [pos, end), reads more behind it
    if (eof) return false;
    if (pos > 0) { memmove(buf.data(), buf.data() + pos, end - pos); end -= pos; pos = 0; }
    if (end == buf.size()) buf.resize(buf.size() * 2);
    const size_t room = std::min<size_t>(buf.size() - end, 1u << 30);
    long n = f ? (long)gzread(f, buf.data() + end, (unsigned)room) : (long)::read(fd, buf.data() + end, room);
    if (n <= 0) { eof = true; return false; }
    end += (size_t)n;
    return true;
  }
  // next line without its terminator (\n or \r\n); false at end of input.  The view stays valid until the next call.
  bool getline(const char*& s, size_t& n) {
    while (true) {
      const char* nl = (const char*)memchr(buf.data() + pos, '\n', end - pos);
      if (nl) {
        s = buf.data() + pos;
        n = (size_t)(nl - s);
        pos += n + 1;
        if (n > 0 && s[n - 1] == '\r') n--;
        line++;
        return true;
      }
      if (!fill()) {
        if (pos < end) { s = buf.data() + pos; n = end - pos; pos = end; if (n > 0 && s[n - 1] == '\r') n--; line++; return true; }
        return false;
      }
    }
  }
  int peek() {  // first byte of the next line, -1 at end of input
    while (pos >= end) if (!fill()) return -1;
    return (unsigned char)buf[pos];
  }
};

struct Record { std::string name, seq, qual; bool hasQual = false; };

void trim(const char*& s, size_t& n) {
  while (n > 0 && (s[0] == ' ' || s[0] == '\t')) { s++; n--; }
  while (n > 0 && (s[n - 1] == ' ' || s[n - 1] == '\t' || s[n - 1] == '\r')) n--;
}

// one FASTA or FASTQ record (cli.read_sequences: the name is the header up to the first blank; FASTA sequences may span lines; a FASTQ record is four lines)
int nextRecord(Source& src, Record& r) {
  const char* s; size_t n;
  while (true) {
    if (!src.getline(s, n)) return 1;
    if (n > 0) break;
  }
  if (s[0] != '>' && s[0] != '@') return fail(src.path + ": line " + std::to_string(src.line) + " is neither a FASTA nor a FASTQ header");
  const bool fastq = s[0] == '@';
  size_t k = 1;
  while (k < n && s[k] != ' ' && s[k] != '\t') k++;
  r.name.assign(s + 1, k - 1);
  r.seq.clear(); r.qual.clear(); r.hasQual = false;
  if (fastq) {
    if (!src.getline(s, n)) return fail(src.path + ": FASTQ record without a sequence line at line " + std::to_string(src.line));
    trim(s, n);
    r.seq.assign(s, n);
    if (src.getline(s, n)) {  // '+'
      if (src.getline(s, n)) { trim(s, n); r.qual.assign(s, n); }
    }
    r.hasQual = true;
  } else {
    while (true) {
      const int c = src.peek();
      if (c < 0 || c == '>') break;
      src.getline(s, n);
      trim(s, n);
      r.seq.append(s, n);
    }
  }
  return 0;
}

}  // namespace

struct xmio_reader {
  Source a, b;
  bool paired = false;
  int split = 0;
  bool keepQual = false;
  // a record that was read and split into sections, of which the current batch took only some
  std::vector<std::pair<int, int>> pendingSections;
  size_t pendingNext = 0;
  Record pending;
};

struct BatchStore {
  std::vector<int32_t> mateCount, mateLength;
  std::vector<int64_t> mateOffset, nameOff, qualOff;
  std::vector<uint8_t> codes;
  std::string names, quals;
  std::vector<uint8_t> hasQual;
};

extern "C" {

const char* xmio_last_error(void) { return g_error.c_str(); }

xmio_reader* xmio_open(const char* path1, const char* path2, int32_t split_past_size, int32_t keep_qualities) {
  xmio_reader* r = new xmio_reader();
  if (!path1 || !r->a.open(path1)) { g_error = std::string("cannot open ") + (path1 ? path1 : "(null)"); delete r; return nullptr; }
  if (path2) {
    if (!r->b.open(path2)) { g_error = std::string("cannot open ") + path2; r->a.close(); delete r; return nullptr; }
    r->paired = true;
  }
  r->split = split_past_size;
  r->keepQual = keep_qualities != 0;
  return r;
}

void xmio_close(xmio_reader* r) {
  if (!r) return;
  r->a.close(); r->b.close();
  delete r;
}

void xmio_batch_free(xmio_batch* b) {
  if (!b) return;
  delete (BatchStore*)b->store;
  delete b;
}

// Up to max_queries queries (a pair is one query) into a new batch; *out = null at end of input.
int xmio_next(xmio_reader* r, int64_t max_queries, xmio_batch** out) {
  *out = nullptr;
  BatchStore* st = new BatchStore();
  auto addMate = [&](const std::string& name, const char* seq, size_t n, const std::string* qual) {
    st->mateOffset.push_back((int64_t)st->codes.size());
    st->mateLength.push_back((int32_t)n);
    const size_t at = st->codes.size();
    st->codes.resize(at + n);
    for (size_t i = 0; i < n; i++) st->codes[at + i] = g_code[(uint8_t)seq[i]];
    st->nameOff.push_back((int64_t)st->names.size());
    st->names += name;
    if (r->keepQual) {
      st->qualOff.push_back((int64_t)st->quals.size());
      st->hasQual.push_back(qual ? 1 : 0);
      if (qual) st->quals += *qual;
    }
  };
  auto padMate = [&]() {  // the unused second mate of a single query
    st->mateOffset.push_back(0); st->mateLength.push_back(0);
    st->nameOff.push_back((int64_t)st->names.size());
    if (r->keepQual) { st->qualOff.push_back((int64_t)st->quals.size()); st->hasQual.push_back(0); }
  };
  int64_t nq = 0;
  Record ra, rb;
  st->codes.reserve((size_t)std::min<int64_t>(max_queries, 1 << 20) * 160);
  // one mate straight from the file buffer into the batch (the common case: no sections); 0 = added, 1 = end of input, -1 = error
  auto directMate = [&](Source& src) -> int {
    const char* s; size_t n;
    while (true) {
      if (!src.getline(s, n)) return 1;
      if (n > 0) break;
    }
    if (s[0] != '>' && s[0] != '@') return fail(src.path + ": line " + std::to_string(src.line) + " is neither a FASTA nor a FASTQ header");
    const bool fastq = s[0] == '@';
    size_t k = 1;
    while (k < n && s[k] != ' ' && s[k] != '\t') k++;
    st->nameOff.push_back((int64_t)st->names.size());
    st->names.append(s + 1, k - 1);
    const size_t at = st->codes.size();
    st->mateOffset.push_back((int64_t)at);
    auto addBases = [&](const char* p, size_t m) {
      const size_t a0 = st->codes.size();
      st->codes.resize(a0 + m);
      uint8_t* c = st->codes.data() + a0;
      for (size_t i = 0; i < m; i++) c[i] = g_code[(uint8_t)p[i]];
    };
    if (fastq) {
      if (!src.getline(s, n)) return fail(src.path + ": FASTQ record without a sequence line at line " + std::to_string(src.line));
      trim(s, n);
      addBases(s, n);
      bool haveQual = false;
      if (src.getline(s, n) && src.getline(s, n)) { trim(s, n); haveQual = true; }
      if (r->keepQual) {
        st->qualOff.push_back((int64_t)st->quals.size());
        st->hasQual.push_back(1);
        if (haveQual) st->quals.append(s, n);
      }
    } else {
      while (true) {
        const int c = src.peek();
        if (c < 0 || c == '>') break;
        src.getline(s, n);
        trim(s, n);
        addBases(s, n);
      }
      if (r->keepQual) { st->qualOff.push_back((int64_t)st->quals.size()); st->hasQual.push_back(0); }
    }
    st->mateLength.push_back((int32_t)(st->codes.size() - at));
    return 0;
  };
  while (r->split <= 0 && nq < max_queries) {
    int rc = directMate(r->a);
    if (rc < 0) { delete st; return -1; }
    if (r->paired) {
      int rc2 = rc == 1 ? nextRecord(r->b, rb) : directMate(r->b);   // (at the end of the first file the second must end too)
      if (rc2 < 0) { delete st; return -1; }
      if (rc != rc2) { delete st; return fail("paired query files have different numbers of reads: " + r->a.path + ", " + r->b.path); }
    }
    if (rc == 1) break;
    if (!r->paired) padMate();
    st->mateCount.push_back(r->paired ? 2 : 1);
    nq++;
  }
  while (r->split > 0 && nq < max_queries) {
    if (r->pendingNext < r->pendingSections.size()) {  // sections of a long read left over from the previous batch
      const auto se = r->pendingSections[r->pendingNext++];
      addMate(r->pending.name, r->pending.seq.data() + se.first, (size_t)(se.second - se.first), nullptr);
      padMate();
      st->mateCount.push_back(1);
      nq++;
      continue;
    }
    int rc = nextRecord(r->a, ra);
    if (rc < 0) { delete st; return -1; }
    if (r->paired) {
      int rc2 = nextRecord(r->b, rb);
      if (rc2 < 0) { delete st; return -1; }
      if (rc != rc2) { delete st; return fail("paired query files have different numbers of reads: " + r->a.path + ", " + r->b.path); }
    }
    if (rc == 1) break;
    if (r->paired) {
      addMate(ra.name, ra.seq.data(), ra.seq.size(), ra.hasQual ? &ra.qual : nullptr);
      addMate(rb.name, rb.seq.data(), rb.seq.size(), rb.hasQual ? &rb.qual : nullptr);
      st->mateCount.push_back(2);
      nq++;
    } else if (r->split > 0) {
      // SequenceSplitter.java:9-38: (length - 1) / max + 1 sections, section k = [length * k / n, length * (k + 1) / n) in 64-bit integers; the sections carry no quality
      const long long length = (long long)ra.seq.size();
      const long long num = length > 0 ? (length - 1) / r->split + 1 : 1;
      r->pendingSections.clear();
      for (long long k = 0; k < num; k++) r->pendingSections.push_back(std::make_pair((int)(length * k / num), (int)(length * (k + 1) / num)));
      r->pendingNext = 0;
      r->pending = ra;
    } else {
      addMate(ra.name, ra.seq.data(), ra.seq.size(), ra.hasQual ? &ra.qual : nullptr);
      padMate();
      st->mateCount.push_back(1);
      nq++;
    }
  }
  if (nq == 0) { delete st; return 0; }
  st->nameOff.push_back((int64_t)st->names.size());
  if (r->keepQual) st->qualOff.push_back((int64_t)st->quals.size());
  if (st->codes.empty()) st->codes.push_back(0);
  xmio_batch* b = new xmio_batch();
  b->store = st;
  b->num_queries = nq;
  b->mate_count = st->mateCount.data(); b->mate_offset = st->mateOffset.data(); b->mate_length = st->mateLength.data();
  b->codes = st->codes.data(); b->codes_length = (int64_t)st->codes.size();
  b->names = st->names.data(); b->name_off = st->nameOff.data();
  b->quals = r->keepQual ? st->quals.data() : nullptr; b->qual_off = r->keepQual ? st->qualOff.data() : nullptr; b->has_qual = r->keepQual ? st->hasQual.data() : nullptr;
  *out = b;
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------- writing
namespace {

// Double.toString as mapper_amd/sam.py java_double has it: the shortest decimal that reads back as the same double, plain notation with at least one
// fractional digit for 1e-3 <= |x| < 1e7, otherwise d.dddE<n>
void javaDouble(double x, std::string& out) {
  if (x != x) { out += "NaN"; return; }
  if (x == 0) { out += "0.0"; return; }  // (sam.py: x == 0 -> "0.0", also for -0.0)
  if (x > 1.7976931348623157e308) { out += "Infinity"; return; }
  if (x < -1.7976931348623157e308) { out += "-Infinity"; return; }
  char buf[64];
  auto res = std::to_chars(buf, buf + sizeof(buf), x, std::chars_format::scientific);  // shortest round trip: d[.ddd]e[+-]xx
  std::string s(buf, res.ptr);
  size_t i = 0;
  bool neg = false;
  if (s[0] == '-') { neg = true; i = 1; }
  const size_t e = s.find('e');
  std::string digits;
  for (size_t k = i; k < e; k++) if (s[k] != '.') digits += s[k];
  const int exp10 = atoi(s.c_str() + e + 1);
  const double a = x < 0 ? -x : x;
  if (neg) out += '-';
  if (a >= 1e-3 && a < 1e7) {
    if (exp10 >= 0) {
      const size_t intDigits = (size_t)exp10 + 1;
      if (digits.size() <= intDigits) { out += digits; out.append(intDigits - digits.size(), '0'); out += ".0"; }
      else { out.append(digits, 0, intDigits); out += '.'; out.append(digits, intDigits, std::string::npos); }
    } else {
      out += "0.";
      out.append((size_t)(-exp10 - 1), '0');
      out += digits;
    }
  } else {
    out += digits[0];
    out += '.';
    if (digits.size() > 1) out.append(digits, 1, std::string::npos); else out += '0';
    out += 'E';
    out += std::to_string(exp10);
  }
}

void appendInt(std::string& out, long long v) {
  char buf[24];
  auto res = std::to_chars(buf, buf + sizeof(buf), v);
  out.append(buf, res.ptr);
}

struct SeqAl { int contig, rev, nb; const int32_t* blocks; };
struct Stats { long long aligned = 0, totalLen = 0, indels = 0; };

void appendCigar(std::string& out, const SeqAl& sa, int queryLen) {  // sam.py cigar()
  const int32_t* first = sa.blocks;
  const int32_t* last = sa.blocks + 4 * (sa.nb - 1);
  if (first[0] > 0) { appendInt(out, first[0]); out += 'S'; }  // [unpinned]
  char prevOp = 0;
  long long prevN = 0;
  for (int k = 0; k < sa.nb; k++) {
    const int32_t la = sa.blocks[4 * k + 2], lb = sa.blocks[4 * k + 3];
    char op; long long n;
    if (la == lb) { op = 'M'; n = la; } else if (lb == 0) { op = 'I'; n = la; } else { op = 'D'; n = lb; }
    if (prevOp == op) prevN += n;
    else { if (prevOp) { appendInt(out, prevN); out += prevOp; } prevOp = op; prevN = n; }
  }
  if (prevOp) { appendInt(out, prevN); out += prevOp; }
  const int tail = queryLen - (last[0] + last[2]);
  if (tail > 0) { appendInt(out, tail); out += 'S'; }  // [unpinned]
}

void appendSeq(std::string& out, const uint8_t* codes, int n, bool rev) {
  const size_t at = out.size();
  out.resize(at + (size_t)n);
  char* p = &out[at];
  if (!rev) for (int i = 0; i < n; i++) p[i] = g_decode[codes[i] & 15];
  else for (int i = 0; i < n; i++) p[i] = g_decode[g_comp[codes[n - 1 - i] & 15]];
}

struct FormatJob {
  const xmio_batch* b;
  const int32_t* ints; const double* dbls; const int64_t* intOff; const int64_t* dblOff;
  const char* const* contigs; int nContigs;
  bool wantSam, wantUnaligned;
};

// the records of queries [q0, q1) appended to sam / unaligned; false = malformed streams
bool formatRange(const FormatJob& J, int64_t q0, int64_t q1, std::string& sam, std::string& un, Stats& st) {
  const xmio_batch* b = J.b;
  for (int64_t q = q0; q < q1; q++) {
    const int32_t* ii = J.ints + J.intOff[q];
    const int32_t* iiEnd = J.ints + J.intOff[q + 1];
    const double* dd = J.dbls + J.dblOff[q];
    const int nMates = b->mate_count[q];
    const bool paired = nMates > 1;
    const int ncomp = *ii++;
    bool any = false;
    for (int c = 0; c < ncomp; c++) {
      const int nal = *ii++;
      for (int a = 0; a < nal; a++) {
        any = true;
        ii++;  // innerDistance
        const int nseq = *ii++;
        const double spacingPenalty = dd[0], penalty = dd[3];
        dd += 4;
        SeqAl sas[2];
        if (nseq < 1 || nseq > 2) return false;
        for (int k = 0; k < nseq; k++) {
          sas[k].contig = ii[0]; sas[k].rev = ii[1]; sas[k].nb = ii[2]; sas[k].blocks = ii + 3;
          ii += 3 + 4 * sas[k].nb;
          dd += 2;
          if (sas[k].nb < 1 || sas[k].contig < 0 || sas[k].contig >= J.nContigs || ii > iiEnd) return false;
          for (int j = 0; j < sas[k].nb; j++) { st.totalLen += sas[k].blocks[4 * j + 2]; if (sas[k].blocks[4 * j + 2] != sas[k].blocks[4 * j + 3]) st.indels++; }
        }
        if (!J.wantSam) continue;
        for (int k = 0; k < nseq; k++) {
          const SeqAl& sa = sas[k];
          const int mate = ncomp == 1 ? k : c;  // (a pair that fell back to unpaired alignments, AlignerWorker.java:602-644: one component per mate)
          if (mate >= nMates) return false;
          const int64_t m = 2 * q + mate;
          sam.append(b->names + b->name_off[m], (size_t)(b->name_off[m + 1] - b->name_off[m]));
          sam += '\t';
          int flag;
          const SeqAl* other = nullptr;
          if (ncomp == 1) {
            if (paired) {
              if (nseq != 2) return false;
              other = &sas[1 - k];
              flag = 1 | 2 | (sa.rev ? 0x10 : 0) | (other->rev ? 0x20 : 0) | (k == 0 ? 0x40 : 0x80);
            } else {
              flag = sa.rev ? 0x10 : 0;  // 16 is [unpinned]
            }
          } else {
            flag = 1 | 8 | (sa.rev ? 0x10 : 0) | (mate == 0 ? 0x40 : 0x80);
          }
          appendInt(sam, flag); sam += '\t';
          sam += J.contigs[sa.contig]; sam += '\t';
          appendInt(sam, (long long)sa.blocks[1] + 1); sam += "\t255\t";
          const int len = b->mate_length[m];
          appendCigar(sam, sa, len); sam += '\t';
          if (other) { sam += J.contigs[other->contig]; sam += '\t'; appendInt(sam, (long long)other->blocks[1] + 1); }
          else sam += "*\t0";
          sam += '\t';
          appendInt(sam, len); sam += '\t';
          appendSeq(sam, b->codes + b->mate_offset[m], len, sa.rev != 0);
          sam += "\t*";
          if (ncomp != 1) sam += "\tcs:f:0.0";
          else if (paired) { sam += "\tcs:f:"; javaDouble(spacingPenalty, sam); }
          sam += "\tAS:f:";
          javaDouble(penalty, sam);
          sam += '\n';
        }
      }
    }
    if (any) st.aligned++;
    else if (J.wantUnaligned) {  // [unpinned format] the query as it came in: FASTQ when it had qualities, else FASTA
      for (int mate = 0; mate < nMates; mate++) {
        const int64_t m = 2 * q + mate;
        const bool fq = b->has_qual && b->has_qual[m];
        un += fq ? '@' : '>';
        un.append(b->names + b->name_off[m], (size_t)(b->name_off[m + 1] - b->name_off[m]));
        un += '\n';
        appendSeq(un, b->codes + b->mate_offset[m], b->mate_length[m], false);
        un += '\n';
        if (fq) { un += "+\n"; un.append(b->quals + b->qual_off[m], (size_t)(b->qual_off[m + 1] - b->qual_off[m])); un += '\n'; }
      }
    }
  }
  return true;
}

bool writeAll(int fd, const std::string& s) {
  size_t done = 0;
  while (done < s.size()) {
    ssize_t n = write(fd, s.data() + done, s.size() - done);
    if (n < 0) return false;
    done += (size_t)n;
  }
  return true;
}

}  // namespace

extern "C" {

// SAM records (no header) of a batch's result streams to sam_fd, its unaligned queries to unaligned_fd (either may be -1), in query order.
int xmio_write_batch(const xmio_batch* b, const int32_t* ints, const double* dbls, const int64_t* int_off, const int64_t* dbl_off, int32_t num_contigs,
                     const char* const* contig_names, int32_t sam_fd, int32_t unaligned_fd, int32_t threads, xmio_stats* stats) {
  if (!b || !ints || !dbls || !int_off || !dbl_off || !stats) return fail("xmio_write_batch: null argument");
  FormatJob J{b, ints, dbls, int_off, dbl_off, contig_names, num_contigs, sam_fd >= 0, unaligned_fd >= 0};
  const int64_t nq = b->num_queries;
  if (threads < 1) threads = 1;
  const int64_t chunk = 4096;
  const int64_t nChunks = (nq + chunk - 1) / chunk;
  // chunks are formatted by the threads in any order and written in order, a window of them at a time (memory stays O(window))
  const int64_t window = std::max<int64_t>((int64_t)threads * 4, 8);
  std::vector<std::string> sam((size_t)window), un((size_t)window);
  std::vector<Stats> st((size_t)nChunks);
  std::atomic<bool> bad(false);
  for (int64_t w0 = 0; w0 < nChunks; w0 += window) {
    const int64_t w1 = std::min(nChunks, w0 + window);
    std::atomic<int64_t> next(w0);
    auto work = [&]() {
      while (true) {
        const int64_t c = next.fetch_add(1);
        if (c >= w1) break;
        std::string& s = sam[(size_t)(c - w0)];
        std::string& u = un[(size_t)(c - w0)];
        s.clear(); u.clear();
        if (!formatRange(J, c * chunk, std::min(nq, (c + 1) * chunk), s, u, st[(size_t)c])) bad = true;
      }
    };
    const int nt = (int)std::min<int64_t>(threads, w1 - w0);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (bad) return fail("xmio_write_batch: malformed result streams");
    for (int64_t c = w0; c < w1; c++) {
      if (sam_fd >= 0 && !writeAll(sam_fd, sam[(size_t)(c - w0)])) return fail("xmio_write_batch: write to the SAM file failed");
      if (unaligned_fd >= 0 && !writeAll(unaligned_fd, un[(size_t)(c - w0)])) return fail("xmio_write_batch: write to the unaligned-query file failed");
    }
  }
  long long aligned = 0, totalLen = 0, indels = 0;
  for (auto& s : st) { aligned += s.aligned; totalLen += s.totalLen; indels += s.indels; }
  // the penalties in query order, as Mapper.run sums them (a double sum: the order is part of the result)
  double totalPenalty = stats->total_penalty;
  for (int64_t q = 0; q < nq; q++) {
    const int32_t* ii = ints + int_off[q];
    const double* dd = dbls + dbl_off[q];
    const int ncomp = *ii++;
    for (int c = 0; c < ncomp; c++) {
      const int nal = *ii++;
      for (int a = 0; a < nal; a++) {
        const int nseq = ii[1];
        ii += 2;
        totalPenalty += dd[3];
        dd += 4;
        for (int k = 0; k < nseq; k++) { ii += 3 + 4 * ii[2]; dd += 2; }
      }
    }
  }
  stats->num_queries += nq;
  stats->num_aligned += aligned;
  stats->total_aligned_length += totalLen;
  stats->num_indels += indels;
  stats->total_penalty = totalPenalty;
  return 0;
}

// Double.toString of x as the SAM tags print it (tests)
int xmio_java_double(double x, char* out, int32_t cap) {
  std::string s;
  javaDouble(x, s);
  if ((int)s.size() + 1 > cap) return -1;
  memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();
}

}  // extern "C"
