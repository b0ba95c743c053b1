// Graph loading: GFA 1 / GFA 2 text and vg protobuf streams, into a forward-only CSR.
//
// Replaces gum::util::load(graph, path, ExternalLoader{parse_vg}, true) as psikt calls it
// (reference src/psikt.cpp:249-251).  gum and libprotobuf are not available; the vg wire
// format is decoded directly (schema: reference vg/vg.proto:13-103; stream framing:
// vg/stream.hpp:81-130 -- gzip, then repeated [varint count, count x (varint len, bytes)],
// a group's first message possibly being the type tag "VG").
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <unordered_map>

#include "host.hpp"

namespace psigpu {
namespace {

struct RawGraph {
  std::vector<uint64_t> ids;
  std::vector<uint64_t> seq_off;                         // into seqbuf
  std::vector<uint32_t> seq_len;
  std::string seqbuf;                                    // labels in file order, upper case
  std::vector<std::pair<uint64_t, uint64_t>> edges;      // external ids, forward
  std::vector<std::pair<std::string, std::vector<uint64_t>>> paths;
  bool follow_reversing = false;                         // PSIGPU_GRAPH_FOLLOW_REVERSING: every link is its `from -> to`, whatever its sides

  // (a node that is defined twice keeps its last definition: resolved when ranks are assigned)
  void add_node(uint64_t id, const char* s, size_t n)
  {
    ids.push_back(id);
    seq_off.push_back(seqbuf.size());
    seq_len.push_back((uint32_t)n);
    const size_t at = seqbuf.size();
    seqbuf.append(s, n);
    for (size_t i = at; i < at + n; ++i) seqbuf[i] = (char)toupper((unsigned char)seqbuf[i]);
  }
  void add_node(uint64_t id, const std::string& s) { add_node(id, s.data(), s.size()); }
};

inline bool parse_u64(const char* p, size_t n, uint64_t* out)
{
  if (n == 0) return false;
  uint64_t x = 0;
  for (size_t i = 0; i < n; ++i) {
    if (p[i] < '0' || p[i] > '9') return false;
    x = x * 10 + (uint64_t)(p[i] - '0');
  }
  *out = x;
  return true;
}

bool strip_orient(const std::string& tok, uint64_t* id, bool* rev)
{
  if (tok.size() < 2) return false;
  char o = tok.back();
  if (o != '+' && o != '-') return false;
  *rev = (o == '-');
  return parse_u64(tok.data(), tok.size() - 1, id);
}

// Orientation handling: the reference traverser follows `to` ids only and ignores the link
// type (include/psi/traverser_bfs.hpp:146-160), i.e. it assumes forward-only graphs.  A
// (a-, b-) link is the forward link (b+, a+); anything else that reverses is rejected.
//
// PSIGPU_GRAPH_FOLLOW_REVERSING restates the reference literally instead: its traverser asks gum for the out-links of a node
// and follows each link's `to` id, reading that node forwards whatever side the link enters it by (`linktype` is discarded,
// traverser_bfs.hpp:146-160) -- so every link, reversing or not, is the edge from -> to as the file writes it, and a reverse
// step of an embedded path is the node itself.  Graphs with inversions load that way; the hit set is the reference's, not
// what a strand-aware mapper would call correct.
bool add_edge(RawGraph& rg, uint64_t a, bool ar, uint64_t b, bool br, std::string* err)
{
  if (rg.follow_reversing) { rg.edges.emplace_back(a, b); return true; }
  if (ar && br) std::swap(a, b);
  else if (ar || br) { *err = "reversing edges are not supported (PSIGPU_GRAPH_FOLLOW_REVERSING / psikt --follow-reversing-edges walks them as the reference does)"; return false; }
  rg.edges.emplace_back(a, b);
  return true;
}

// A line's tab-separated fields as (pointer, length) pairs into the file buffer.
struct Fields {
  const char* p[8];
  size_t n[8];
  int count = 0;
  void cut(const char* b, const char* e, int want)
  {
    count = 0;
    while (count < want) {
      const char* t = (const char*)memchr(b, '\t', e - b);
      p[count] = b; n[count] = (t && count + 1 < want ? t : e) - b;
      ++count;
      if (!t || count == want) break;
      b = t + 1;
    }
  }
};

bool parse_gfa(const std::string& path, RawGraph& rg, std::string* err)
{
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { *err = "cannot open " + path; return false; }
  std::string buf;
  {
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(sz > 0 ? (size_t)sz : 0);
    size_t got = buf.empty() ? 0 : fread(&buf[0], 1, buf.size(), f);
    fclose(f);
    if (got != buf.size()) { *err = "cannot read " + path; return false; }
  }
  if (buf.size() >= 2 && (unsigned char)buf[0] == 0x1F && (unsigned char)buf[1] == 0x8B) {
    *err = path + " is gzip-compressed: decompress it first (GFA is read as plain text)";
    return false;
  }
  rg.seqbuf.reserve(buf.size() / 2);
  const char* p = buf.data();
  const char* const end = p + buf.size();
  Fields fl;
  while (p < end) {
    const char* nl = (const char*)memchr(p, '\n', end - p);
    const char* e = nl ? nl : end;
    const char* next = nl ? nl + 1 : end;
    if (e > p && e[-1] == '\r') --e;
    if (e - p < 2 || p[1] != '\t') { p = next; continue; }
    const char t = p[0];
    if (t == 'S') {
      fl.cut(p, e, 5);            // S id seq [tags...]  |  GFA 2: S id len seq [tags...]
      if (fl.count < 3) { *err = "bad S line"; return false; }
      uint64_t id, len2;
      // (gum addresses nodes by integer ids, as vg does; a named segment cannot be mapped to one)
      if (!parse_u64(fl.p[1], fl.n[1], &id)) { *err = "non-numeric segment id '" + std::string(fl.p[1], fl.n[1]) + "'"; return false; }
      // the 5th "field" is the rest of the line: cut the 4th at its own tab
      size_t n3 = fl.count >= 4 ? fl.n[3] : 0;
      if (fl.count >= 4 && parse_u64(fl.p[2], fl.n[2], &len2) && memchr(fl.p[3], ':', n3) == nullptr)
        rg.add_node(id, fl.p[3], n3);
      else
        rg.add_node(id, fl.p[2], fl.n[2]);
    } else if (t == 'L') {
      fl.cut(p, e, 6);
      if (fl.count < 5) { *err = "bad L line"; return false; }
      uint64_t a, b;
      if (!parse_u64(fl.p[1], fl.n[1], &a) || !parse_u64(fl.p[3], fl.n[3], &b)) { *err = "non-numeric segment id on an L line"; return false; }
      if (!add_edge(rg, a, fl.n[2] == 1 && fl.p[2][0] == '-', b, fl.n[4] == 1 && fl.p[4][0] == '-', err)) return false;
    } else if (t == 'E') {
      fl.cut(p, e, 5);
      if (fl.count < 4) { *err = "bad E line"; return false; }
      uint64_t a, b; bool ar, br;
      if (!strip_orient(std::string(fl.p[2], fl.n[2]), &a, &ar) || !strip_orient(std::string(fl.p[3], fl.n[3]), &b, &br)) { *err = "bad E line"; return false; }
      if (!add_edge(rg, a, ar, b, br, err)) return false;
    } else if (t == 'O' || t == 'P') {
      fl.cut(p, e, 4);
      if (fl.count < 3) { *err = "bad path line"; return false; }
      const char sep = t == 'O' ? ' ' : ',';
      std::vector<uint64_t> nodes;
      const char* q = fl.p[2];
      const char* const qe = q + fl.n[2];
      while (q < qe) {
        const char* s2 = (const char*)memchr(q, sep, qe - q);
        const char* te = s2 ? s2 : qe;
        if (te > q) {
          const char o = te[-1];
          uint64_t id;
          if ((o != '+' && o != '-') || te - q < 2 || !parse_u64(q, te - q - 1, &id)) { *err = "bad path step"; return false; }
          if (o == '-' && !rg.follow_reversing) { *err = "reverse path steps are not supported"; return false; }
          nodes.push_back(id);
        }
        q = s2 ? s2 + 1 : qe;
      }
      rg.paths.emplace_back(std::string(fl.p[1], fl.n[1]), std::move(nodes));
    }
    p = next;
  }
  return true;
}

// --- minimal protobuf wire decoding -------------------------------------------------
struct PB {
  const uint8_t* p;
  const uint8_t* e;
  bool ok = true;
  uint64_t varint()
  {
    uint64_t x = 0; int s = 0;
    while (p < e) {
      uint8_t b = *p++;
      x |= (uint64_t)(b & 0x7F) << s;
      if (!(b & 0x80)) return x;
      s += 7;
      if (s > 63) break;
    }
    ok = false;
    return 0;
  }
  // returns false at end; for wt==2 sets [sub, sub+len)
  bool next(uint32_t* fno, uint32_t* wt, uint64_t* val, const uint8_t** sub, uint64_t* len)
  {
    if (p >= e || !ok) return false;
    uint64_t key = varint();
    if (!ok) return false;
    *fno = (uint32_t)(key >> 3); *wt = (uint32_t)(key & 7);
    switch (*wt) {
      case 0: *val = varint(); break;
      case 1: if (e - p < 8) { ok = false; return false; } p += 8; break;
      case 5: if (e - p < 4) { ok = false; return false; } p += 4; break;
      case 2: {
        uint64_t l = varint();
        if (!ok || (uint64_t)(e - p) < l) { ok = false; return false; }
        *sub = p; *len = l; p += l;
        break;
      }
      default: ok = false; return false;
    }
    return ok;
  }
};

bool parse_vg_graph_msg(const uint8_t* b, uint64_t n, RawGraph& rg,
                        std::map<std::string, std::vector<std::pair<uint64_t, uint64_t>>>& paths,
                        std::vector<std::string>& path_order, std::string* err)
{
  PB m{ b, b + n };
  uint32_t fno, wt; uint64_t val = 0, len = 0; const uint8_t* sub = nullptr;
  while (m.next(&fno, &wt, &val, &sub, &len)) {
    if (wt != 2) continue;
    if (fno == 1) {                       // Node { sequence = 1; name = 2; id = 3 }
      PB q{ sub, sub + len };
      std::string s; uint64_t id = 0;
      uint32_t a, w; uint64_t v = 0, l = 0; const uint8_t* sp = nullptr;
      while (q.next(&a, &w, &v, &sp, &l)) {
        if (a == 1 && w == 2) s.assign((const char*)sp, l);
        else if (a == 3 && w == 0) id = v;
      }
      if (!q.ok) { *err = "bad vg Node"; return false; }
      rg.add_node(id, s);
    } else if (fno == 2) {                // Edge { from = 1; to = 2; from_start = 3; to_end = 4 }
      PB q{ sub, sub + len };
      uint64_t d[5] = { 0, 0, 0, 0, 0 };
      uint32_t a, w; uint64_t v = 0, l = 0; const uint8_t* sp = nullptr;
      while (q.next(&a, &w, &v, &sp, &l))
        if (w == 0 && a >= 1 && a <= 4) d[a] = v;
      if (!q.ok) { *err = "bad vg Edge"; return false; }
      if (!add_edge(rg, d[1], d[3] != 0, d[2], d[4] != 0, err)) return false;
    } else if (fno == 3) {                // Path { name = 1; mapping = 2 }
      PB q{ sub, sub + len };
      std::string name;
      std::vector<std::pair<uint64_t, uint64_t>> maps;
      uint32_t a, w; uint64_t v = 0, l = 0; const uint8_t* sp = nullptr;
      while (q.next(&a, &w, &v, &sp, &l)) {
        if (a == 1 && w == 2) name.assign((const char*)sp, l);
        else if (a == 2 && w == 2) {      // Mapping { position = 1; edit = 2; rank = 5 }
          PB mq{ sp, sp + l };
          uint64_t nid = 0, rank = 0;
          uint32_t c, cw; uint64_t cv = 0, cl = 0; const uint8_t* cp = nullptr;
          while (mq.next(&c, &cw, &cv, &cp, &cl)) {
            if (c == 1 && cw == 2) {      // Position { node_id = 1; offset = 2; is_reverse = 4 }
              PB pq{ cp, cp + cl };
              uint32_t e, ew; uint64_t ev = 0, el = 0; const uint8_t* ep = nullptr;
              while (pq.next(&e, &ew, &ev, &ep, &el)) {
                if (e == 1 && ew == 0) nid = ev;
                else if (e == 4 && ew == 0 && ev && !rg.follow_reversing) { *err = "reverse path steps are not supported"; return false; }
              }
            } else if (c == 5 && cw == 0) rank = cv;
          }
          maps.emplace_back(rank, nid);
        }
      }
      if (!q.ok) { *err = "bad vg Path"; return false; }
      if (!paths.count(name)) path_order.push_back(name);
      auto& dst = paths[name];
      dst.insert(dst.end(), maps.begin(), maps.end());
    }
  }
  // a message that fails to parse as a Graph (the "VG" tag) is tolerated, as the reference
  // does (vg/stream.hpp:104-114)
  return true;
}

bool parse_vg(const std::string& path, RawGraph& rg, std::string* err)
{
  gzFile gz = gzopen(path.c_str(), "rb");
  if (!gz) { *err = "cannot open " + path; return false; }
  std::vector<uint8_t> raw;
  std::vector<uint8_t> buf(1 << 20);
  int got;
  while ((got = gzread(gz, buf.data(), (unsigned)buf.size())) > 0)
    raw.insert(raw.end(), buf.begin(), buf.begin() + got);
  gzclose(gz);
  if (got < 0) { *err = "gzip error in " + path; return false; }
  std::map<std::string, std::vector<std::pair<uint64_t, uint64_t>>> paths;
  std::vector<std::string> order;
  PB s{ raw.data(), raw.data() + raw.size() };
  while (s.p < s.e) {
    uint64_t cnt = s.varint();
    if (!s.ok) { *err = "bad vg stream"; return false; }
    for (uint64_t i = 0; i < cnt; ++i) {
      uint64_t len = s.varint();
      if (!s.ok || (uint64_t)(s.e - s.p) < len) { *err = "bad vg stream"; return false; }
      const uint8_t* msg = s.p;
      s.p += len;
      if (len == 2 && msg[0] == 'V' && msg[1] == 'G') continue;
      if (!parse_vg_graph_msg(msg, len, rg, paths, order, err)) return false;
    }
  }
  for (auto& name : order) {
    auto& maps = paths[name];
    std::stable_sort(maps.begin(), maps.end(),
                     [](const auto& a, const auto& b) { return a.first < b.first; });
    std::vector<uint64_t> nodes;
    for (auto& m : maps) nodes.push_back(m.second);
    rg.paths.emplace_back(name, std::move(nodes));
  }
  return true;
}

bool ends_with(const std::string& s, const char* suf)
{
  size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

}  // namespace

Graph* load_graph_file(const std::string& path, int* status, std::string* err, uint32_t flags)
{
  RawGraph rg;
  rg.follow_reversing = (flags & PSIGPU_GRAPH_FOLLOW_REVERSING) != 0;
  bool ok;
  if (ends_with(path, ".vg")) ok = parse_vg(path, rg, err);
  else ok = parse_gfa(path, rg, err);
  if (!ok) {
    *status = (err->find("cannot open") == 0) ? PSIGPU_ERR_IO : PSIGPU_ERR_FORMAT;
    return nullptr;
  }
  if (rg.ids.size() >= 0xFFFFFFF0ull) { *status = PSIGPU_ERR_FORMAT; *err = "too many nodes"; return nullptr; }
  Graph* g = new Graph;
  // Node rank = position in ascending external-id order (psikt loads with sort = true,
  // src/psikt.cpp:249-251), so .gfa and .vg renderings of one graph give identical ranks.  A node
  // defined twice keeps its last definition.
  const uint64_t n_def = rg.ids.size();
  std::vector<uint32_t> perm(n_def);
  for (uint64_t i = 0; i < n_def; ++i) perm[i] = (uint32_t)i;
  bool sorted = true;
  for (uint64_t i = 1; i < n_def && sorted; ++i) sorted = rg.ids[i - 1] < rg.ids[i];
  if (!sorted) std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return rg.ids[a] < rg.ids[b]; });
  g->node_id.reserve(n_def);
  g->label_off.assign(1, 0);
  g->label_off.reserve(n_def + 1);
  g->labels.reserve(rg.seqbuf.size());
  for (uint64_t i = 0; i < n_def; ++i) {
    const uint32_t d = perm[i];
    if (i + 1 < n_def && rg.ids[perm[i + 1]] == rg.ids[d]) continue;      // a later definition follows
    g->node_id.push_back(rg.ids[d]);
    g->labels.append(rg.seqbuf, rg.seq_off[d], rg.seq_len[d]);
    g->label_off.push_back(g->labels.size());
  }
  const uint64_t n = g->node_id.size();
  // id -> rank: a direct table when the ids are dense (the usual case), bisection otherwise
  const uint64_t id_min = n ? g->node_id.front() : 0, id_max = n ? g->node_id.back() : 0;
  const bool dense = n && id_max - id_min < 4 * n + 1024;
  std::vector<uint32_t> direct;
  if (dense) {
    direct.assign(id_max - id_min + 1, 0xFFFFFFFFu);
    for (uint64_t r = 0; r < n; ++r) direct[g->node_id[r] - id_min] = (uint32_t)r;
  }
  auto rank_of = [&](uint64_t id, uint32_t* r) -> bool {
    if (dense) {
      if (id < id_min || id > id_max || direct[id - id_min] == 0xFFFFFFFFu) return false;
      *r = direct[id - id_min];
      return true;
    }
    auto it = std::lower_bound(g->node_id.begin(), g->node_id.end(), id);
    if (it == g->node_id.end() || *it != id) return false;
    *r = (uint32_t)(it - g->node_id.begin());
    return true;
  };
  // CSR in edge file order per source node, duplicates dropped: count, place, then squeeze
  std::vector<uint32_t> ea(rg.edges.size()), eb(rg.edges.size());
  g->edge_off.assign(n + 1, 0);
  for (size_t i = 0; i < rg.edges.size(); ++i) {
    if (!rank_of(rg.edges[i].first, &ea[i]) || !rank_of(rg.edges[i].second, &eb[i])) {
      *status = PSIGPU_ERR_FORMAT; *err = "edge refers to an unknown node"; delete g; return nullptr;
    }
    ++g->edge_off[ea[i] + 1];
  }
  for (uint64_t v = 0; v < n; ++v) g->edge_off[v + 1] += g->edge_off[v];
  {
    std::vector<uint32_t> to(rg.edges.size());
    std::vector<uint64_t> fill(g->edge_off.begin(), g->edge_off.end() - 1);
    for (size_t i = 0; i < rg.edges.size(); ++i) to[fill[ea[i]]++] = eb[i];       // stable: file order kept
    g->edge_to.reserve(to.size());
    uint64_t prev_end = 0;
    for (uint64_t v = 0; v < n; ++v) {
      const uint64_t b0 = prev_end, e0 = g->edge_off[v + 1];
      const uint64_t first = g->edge_to.size();
      for (uint64_t j = b0; j < e0; ++j) {
        bool dup = false;
        for (uint64_t q = first; q < g->edge_to.size() && !dup; ++q) dup = g->edge_to[q] == to[j];
        if (!dup) g->edge_to.push_back(to[j]);
      }
      prev_end = e0;
      g->edge_off[v] = first;
    }
    g->edge_off[n] = g->edge_to.size();
  }
  for (auto& p : rg.paths) {
    std::vector<uint32_t> nodes;
    nodes.reserve(p.second.size());
    for (uint64_t id : p.second) {
      uint32_t r;
      if (!rank_of(id, &r)) {
        *status = PSIGPU_ERR_FORMAT; *err = "path refers to an unknown node"; delete g; return nullptr;
      }
      nodes.push_back(r);
    }
    g->paths.push_back(std::move(nodes));
    g->path_names.push_back(p.first);
  }
  *status = PSIGPU_OK;
  return g;
}

}  // namespace psigpu
