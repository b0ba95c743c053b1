// Reading the reference's own path-set file `<prefix>_paths`.
//
// Written by PathIndex::save_paths_set (reference include/psi/pathindex.hpp:315-332):
//     u64 context | u64 direction (1 = Forward, 0 = Reversed) | PathSet::serialize (pathset.hpp:260-274):
//     u64 #paths | per path Path< Graph, Compact >::serialize (path_base.hpp:551-560):
//         sdsl::enc_vector< coder::elias_delta<> > of the node ids in coordinate (= external) ids (:701-710),
//         u64 left, u64 right (bases of the first / last node that belong to the path, 0 = all: :113-114,
//         :382-438), sdsl::bit_vector of node breaks (one bit per base of the path sequence, :663-678)
//     | the FM index over the ",id,id,...," strings (pathset.hpp:311-321) | one bit_vector per path (:323-349)
// Only the paths and their trims are needed: the FM index over the path sequences is rebuilt here from
// them (psigpu_index_from_reference_paths), so reading stops behind the last path.  The companion
// `<prefix>` file (sdsl::csa_wt of the reference's reversed text) is not read at all.
//
// sdsl-lite is not in the reference tree; its serialisation is restated from the published sources
// (sdsl-lite 2.1.1: int_vector.hpp `serialize` / `int_vector_trait<0>::write_header`, enc_vector.hpp
// `serialize` and constructor, coder_elias_delta.hpp):
//     int_vector<0>:  u64 size in BITS, u8 width, ceil(bits / 64) u64 words (entry i at bit i * width, LSB first)
//     bit_vector:     u64 size in bits, ceil(bits / 64) u64 words
//     enc_vector<elias_delta, 128, 0>:  u64 size | int_vector<0> z (width 1: the code bits)
//                                       | int_vector<0> samples_and_pointers
//         every 128th value is a sample, kept verbatim beside the bit offset of its run in z; the other values
//         are v[i] = v[i-1] + delta (mod 2^64) with Elias-delta coded deltas:
//         ll zero bits, a one, the low ll bits of len, the low len - 1 bits of delta  (len = bits of delta;
//         a delta of 0 is coded as 2^64: len 65)
// What cannot be verified here: gum's id / offset widths never reach this file (ids are enc_vector values,
// the trims explicit u64), but an sdsl version that lays int_vector out differently would not be read.
#include <cstdio>
#include <sys/types.h>
#include <cstring>
#include <unordered_map>

#include "host.hpp"

namespace psigpu {
namespace {

// The file is untrusted: every size field is checked against the bytes the file still holds BEFORE anything is
// allocated for it (a garbage header must not ask for 128 GiB), and the handle closes itself on every way out.
struct Reader {
  FILE* f = nullptr;
  bool ok = true;
  uint64_t left = 0;                            // bytes of the file not read yet
  explicit Reader(const char* path) : f(fopen(path, "rb"))
  {
    if (!f) { ok = false; return; }
    if (fseeko(f, 0, SEEK_END) == 0) { const off_t e = ftello(f); if (e > 0) left = (uint64_t)e; }
    if (fseeko(f, 0, SEEK_SET) != 0) ok = false;
  }
  ~Reader() { if (f) fclose(f); }
  Reader(const Reader&) = delete;
  Reader& operator=(const Reader&) = delete;
  bool take(void* dst, uint64_t bytes)
  {
    if (!ok || bytes > left || (bytes && fread(dst, 1, bytes, f) != bytes)) return ok = false;
    left -= bytes;
    return true;
  }
  uint64_t u64() { uint64_t v = 0; take(&v, 8); return ok ? v : 0; }
  uint8_t u8() { uint8_t v = 0; take(&v, 1); return ok ? v : 0; }
  // `bits` bits of payload as u64 words
  bool words(uint64_t bits, std::vector<uint64_t>& out)
  {
    if (!ok || bits > left * 8) return ok = false;          // (left < 2^61: no overflow)
    const uint64_t nw = (bits + 63) / 64;
    if (nw * 8 > left) return ok = false;
    out.assign(nw + 1, 0);                      // (+1: reads that straddle the last word stay inside)
    return take(out.data(), nw * 8);
  }
};

struct IntVector { uint64_t bits = 0; uint8_t width = 0; std::vector<uint64_t> w; };

inline uint64_t get_bits(const std::vector<uint64_t>& w, uint64_t at, uint32_t n)       // n <= 64
{
  if (n == 0) return 0;
  const uint64_t i = at >> 6, sh = at & 63;
  uint64_t x = w[i] >> sh;
  if (sh + n > 64) x |= w[i + 1] << (64 - sh);
  return n == 64 ? x : (x & ((1ull << n) - 1));
}

bool read_int_vector0(Reader& r, IntVector& v)
{
  v.bits = r.u64();
  v.width = r.u8();
  return r.ok && v.width <= 64 && r.words(v.bits, v.w);
}

// one Elias-delta code at bit `at` of z (at most z_bits bits): advances `at`
bool elias_delta_decode(const std::vector<uint64_t>& z, uint64_t z_bits, uint64_t& at, uint64_t* x)
{
  uint32_t ll = 0;
  while (true) {
    if (at >= z_bits) return false;
    if (get_bits(z, at++, 1)) break;
    if (++ll > 6) return false;                 // len <= 65 < 2^7
  }
  if (ll == 0) { *x = 1; return true; }
  if (at + ll > z_bits) return false;
  const uint32_t len = (1u << ll) | (uint32_t)get_bits(z, at, ll);
  at += ll;
  if (len > 65 || at + (len - 1) > z_bits) return false;
  const uint64_t low = get_bits(z, at, len - 1);
  at += len - 1;
  *x = len == 65 ? 0 : ((1ull << (len - 1)) | low);
  return true;
}

bool read_enc_vector(Reader& r, std::vector<uint64_t>& out)
{
  const uint64_t size = r.u64();
  IntVector z, sp;
  // (a value costs at least one code bit or a share of a sample: `size` beyond the bits left is garbage)
  if (!r.ok || size > r.left * 8 || !read_int_vector0(r, z) || !read_int_vector0(r, sp)) return false;
  out.clear();
  if (size == 0) return true;
  constexpr uint64_t DENS = 128;
  const uint64_t n_samples = (size + DENS - 1) / DENS;
  if (z.width != 1 || sp.width == 0 || sp.bits / sp.width < 2 * n_samples) return false;
  out.reserve(size);
  for (uint64_t s = 0; s < n_samples; ++s) {
    uint64_t v = get_bits(sp.w, (2 * s) * sp.width, sp.width);
    uint64_t at = get_bits(sp.w, (2 * s + 1) * sp.width, sp.width);
    out.push_back(v);
    for (uint64_t i = s * DENS + 1; i < std::min(size, (s + 1) * DENS); ++i) {
      uint64_t d;
      if (!elias_delta_decode(z.w, z.bits, at, &d)) return false;
      v += d;
      out.push_back(v);
    }
  }
  return true;
}

}  // namespace

// `<prefix>_paths` -> the indexed paths as node ranks with their trims in this library's terms (head offset
// into the first node, indexed bases of the last node; 0 = the whole node).
int read_reference_paths(const std::string& file, const Graph& g, uint64_t* context, bool* forward,
                         std::vector<std::vector<uint32_t>>& paths, std::vector<uint32_t>& head,
                         std::vector<uint32_t>& tail, std::string* err)
{
  Reader r(file.c_str());
  if (!r.f) { *err = "cannot open " + file; return PSIGPU_ERR_IO; }
  auto fail = [&](const std::string& why) { *err = file + ": " + why; return PSIGPU_ERR_FORMAT; };
  *context = r.u64();
  const uint64_t dir = r.u64(), n_paths = r.u64();
  // (a path costs at least 8 + 9 + 9 + 8 + 8 + 8 bytes of headers)
  if (!r.ok || dir > 1 || n_paths > (1ull << 32) || n_paths > r.left / 50) return fail("not a path-set file");
  *forward = dir == 1;
  std::unordered_map<uint64_t, uint32_t> rank;
  rank.reserve(g.n_nodes() * 2);
  for (uint64_t v = 0; v < g.n_nodes(); ++v) rank.emplace(g.node_id[v], (uint32_t)v);
  paths.clear(); head.clear(); tail.clear();
  std::vector<uint64_t> ids, bv;
  for (uint64_t p = 0; p < n_paths; ++p) {
    if (!read_enc_vector(r, ids)) return fail("bad enc_vector in path " + std::to_string(p));
    const uint64_t left = r.u64(), right = r.u64(), bv_bits = r.u64();
    if (!r.ok || !r.words(bv_bits, bv)) return fail("truncated in path " + std::to_string(p));
    std::vector<uint32_t> nodes;
    nodes.reserve(ids.size());
    for (uint64_t id : ids) {
      auto it = rank.find(id);
      if (it == rank.end()) return fail("node " + std::to_string(id) + " is not in this graph");
      nodes.push_back(it->second);
    }
    uint32_t h = 0, t = 0;
    uint64_t seqlen = 0;
    if (!nodes.empty()) {
      // left / right: LENGTHS of the first / last node's part of the path (path_base.hpp:113-114); a value of 0
      // or beyond the node means the whole node (set_left_by_len / set_right_by_len, :382-438)
      const uint64_t l0 = g.node_len(nodes.front()), l1 = g.node_len(nodes.back());
      if (left && left < l0) h = (uint32_t)(l0 - left);
      if (right && right < l1) t = (uint32_t)right;
      for (uint32_t v : nodes) seqlen += g.node_len(v);
      seqlen -= h;
      if (t) seqlen -= l1 - t;
      if (nodes.size() == 1 && t && h >= t) return fail("left exceeds right on a one-node path");
      // the node-break bit vector: one bit per base of the path sequence, set at the last base of every node
      uint64_t ones = 0;
      for (size_t i = 0; i + 1 < bv.size(); ++i) ones += (uint64_t)__builtin_popcountll(bv[i]);
      if (bv_bits != seqlen || ones != nodes.size()) return fail("node breaks of path " + std::to_string(p) + " do not fit its nodes in this graph");
    } else if (bv_bits != 0) return fail("node breaks of an empty path");
    paths.push_back(std::move(nodes));
    head.push_back(h); tail.push_back(t);
  }
  return PSIGPU_OK;
}

}  // namespace psigpu
