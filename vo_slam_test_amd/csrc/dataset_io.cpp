// dataset_io.cpp -- the I/O contract of the reference's harness (test/vo_run.cpp) and its vocabulary file, as plain
// host C++ behind the C-ABI (no OpenCV / DBoW3 in this image: PNG decoding is zlib + the five scan-line filters):
//   associate.txt          vo_run.cpp:24-58   (`fin >> rgb_time >> rgb_file >> depth_time >> depth_file`, data_num records)
//   cv::imread(rgb, 1) / cv::imread(depth, -1)  :108-109  8-bit colour as B, G, R / 16-bit depth as stored
//   trajectory files       :154-232   `timestamp tx ty tz qx qy qz qw`, Eigen's default stream format
//   tracking-time report   :138-151   median = sorted[tracked / 2], mean = total / tracked
//   DBoW3::Vocabulary(path) :87       binary (.bin / .dbow3, DBoW3 0.0.1 Vocabulary::toStream layout) or the ORB-SLAM2
//                                     text format -> flat tree arrays -> vo_vocab_create
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/vo_hip.h"

namespace vo {
void set_error(const char *fmt, ...);
}

struct vo_dataset {
  std::vector<std::string> rgb_time, rgb_path, depth_time, depth_path;
};

namespace {

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct Png {
  int w = 0, h = 0, depth = 0, color = 0, channels = 0;
  std::vector<uint8_t> pixels;  // h * w * channels * (depth / 8), 16-bit samples big-endian as in the file
  std::vector<uint8_t> palette;
};

int png_decode(const char *path, Png &P, bool header_only) {
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    vo::set_error("cannot open %s", path);
    return VO_ERR_INVALID;
  }
  std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (buf.size() < 33 || memcmp(buf.data(), sig, 8) != 0) {
    vo::set_error("%s is not a PNG file", path);
    return VO_ERR_INVALID;
  }
  std::vector<uint8_t> idat;
  size_t pos = 8;
  int interlace = 0;
  while (pos + 12 <= buf.size()) {
    const uint32_t len = be32(&buf[pos]);
    const char *type = reinterpret_cast<const char *>(&buf[pos + 4]);
    if (pos + 12 + len > buf.size()) break;
    const uint8_t *d = &buf[pos + 8];
    if (!memcmp(type, "IHDR", 4)) {
      P.w = (int)be32(d), P.h = (int)be32(d + 4), P.depth = d[8], P.color = d[9], interlace = d[12];
      P.channels = P.color == 0 ? 1 : P.color == 2 ? 3 : P.color == 3 ? 1 : P.color == 4 ? 2 : 4;
      if (header_only) return VO_OK;
    } else if (!memcmp(type, "PLTE", 4)) {
      P.palette.assign(d, d + len);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), d, d + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + len;
  }
  const bool packed = P.depth < 8 && (P.color == 0 || P.color == 3);  // 1 / 2 / 4-bit grey or palette indices
  if (P.w <= 0 || P.h <= 0 || interlace != 0 || (!packed && P.depth != 8 && P.depth != 16) ||
      (packed && P.depth != 1 && P.depth != 2 && P.depth != 4) || (P.color == 3 && P.depth > 8)) {
    vo::set_error("%s: unsupported PNG (interlaced, or bit depth %d)", path, P.depth);
    return VO_ERR_INVALID;
  }
  const int bpp = packed ? 1 : P.channels * P.depth / 8;
  const size_t stride = packed ? ((size_t)P.w * P.depth + 7) / 8 : (size_t)P.w * bpp;
  std::vector<uint8_t> raw((stride + 1) * P.h);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) {
    vo::set_error("%s: corrupt PNG data stream", path);
    return VO_ERR_INVALID;
  }
  P.pixels.assign(stride * P.h, 0);
  std::vector<uint8_t> zero(stride, 0);
  for (int y = 0; y < P.h; y++) {
    const uint8_t ft = raw[(stride + 1) * y];
    const uint8_t *in = &raw[(stride + 1) * y + 1];
    uint8_t *cur = &P.pixels[stride * y];
    const uint8_t *up = y ? cur - stride : zero.data();
    for (size_t x = 0; x < stride; x++) {
      const int a = x >= (size_t)bpp ? cur[x - bpp] : 0, b = up[x], c = x >= (size_t)bpp ? up[x - bpp] : 0;
      int pred = 0;
      switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: {
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default:
          vo::set_error("%s: bad PNG filter type %d", path, ft);
          return VO_ERR_INVALID;
      }
      cur[x] = (uint8_t)(in[x] + pred);
    }
  }
  if (packed) {  // unpack to one byte per sample (most significant bits first); grey is scaled to 0..255 like libpng's expansion
    std::vector<uint8_t> un((size_t)P.w * P.h);
    const int maxv = (1 << P.depth) - 1;
    for (int y = 0; y < P.h; y++)
      for (int x = 0; x < P.w; x++) {
        const size_t bit = (size_t)x * P.depth;
        const int v = (P.pixels[stride * y + bit / 8] >> (8 - P.depth - (int)(bit % 8))) & maxv;
        un[(size_t)y * P.w + x] = (uint8_t)(P.color == 0 ? v * 255 / maxv : v);
      }
    P.pixels.swap(un);
    P.depth = 8;
  }
  return VO_OK;
}

// `os << eigen_row_vector` with Eigen's default IOFormat: every coefficient printed with the stream's default
// precision (6 significant digits, %g), padded on the left to the widest one, separated by one space
std::string eigen_row(const double *v, int n) {
  std::vector<std::string> s(n);
  size_t w = 0;
  for (int i = 0; i < n; i++) {
    std::ostringstream o;
    o << v[i];
    s[i] = o.str();
    w = std::max(w, s[i].size());
  }
  std::string out;
  for (int i = 0; i < n; i++) {
    if (i) out += " ";
    out += std::string(w - s[i].size(), ' ') + s[i];
  }
  return out;
}

struct VocNode {
  uint32_t id = 0, parent = 0, word_id = 0;
  double weight = 0;
  uint8_t desc[32];
  std::vector<uint32_t> children;
};


// ---- DBoW3 vocabulary as written through cv::FileStorage (Vocabulary::save(cv::FileStorage&), what DBoW3::Vocabulary(path)
// falls back to for .yml / .yml.gz files; the library is not vendored under the reference: layout restated from its
// published source).  "%YAML:1.0", a mapping `vocabulary:` with k, L, scoringType, weightingType, a sequence `nodes:` of
// flow mappings { nodeId, parentId, weight, descriptor:"dbw3 <type> <cols> b0 b1 ..." } (DBoW2 files: 32 plain numbers)
// in the writer's order -- children are attached to their parent in FILE order, as Vocabulary::load does -- and a
// sequence `words:` of { wordId, nodeId }.  The reader below is a tolerant scanner of exactly that subset (flow mappings
// may wrap over lines), not a YAML parser; gzip is handled by zlib (gzread also passes plain files through).
bool read_all_gz(const char *path, std::string &text) {
  gzFile g = gzopen(path, "rb");
  if (!g) return false;
  char buf[1 << 16];
  int n;
  while ((n = gzread(g, buf, sizeof(buf))) > 0) text.append(buf, (size_t)n);
  gzclose(g);
  return n == 0;
}

struct YamlMap {
  std::vector<std::pair<std::string, std::string>> kv;
  const std::string *get(const char *k) const {
    for (const auto &e : kv)
      if (e.first == k) return &e.second;
    return nullptr;
  }
};

// the flow mappings "{ key:value, key:"string", ... }" between two offsets of the text
bool yaml_flow_maps(const std::string &t, size_t from, size_t to, std::vector<YamlMap> &out) {
  size_t i = from;
  while (true) {
    i = t.find('{', i);
    if (i == std::string::npos || i >= to) return true;
    YamlMap m;
    i++;
    while (i < to) {
      while (i < to && (isspace((unsigned char)t[i]) || t[i] == ',')) i++;
      if (i < to && t[i] == '}') break;
      size_t ks = i;
      while (i < to && t[i] != ':' && t[i] != '}') i++;
      if (i >= to || t[i] != ':') return false;
      std::string key = t.substr(ks, i - ks);
      while (!key.empty() && isspace((unsigned char)key.back())) key.pop_back();
      i++;
      while (i < to && isspace((unsigned char)t[i])) i++;
      std::string val;
      if (i < to && t[i] == '"') {
        size_t e = t.find('"', i + 1);
        if (e == std::string::npos || e >= to) return false;
        val = t.substr(i + 1, e - i - 1);
        i = e + 1;
      } else {
        size_t vs = i;
        while (i < to && t[i] != ',' && t[i] != '}' && !isspace((unsigned char)t[i])) i++;
        val = t.substr(vs, i - vs);
      }
      m.kv.push_back({key, val});
    }
    if (i >= to) return false;
    i++;  // '}'
    out.push_back(std::move(m));
  }
}

bool yaml_scalar(const std::string &t, size_t from, size_t to, const char *key, long &v) {
  const std::string k = std::string(key) + ":";
  for (size_t i = t.find(k, from); i != std::string::npos && i < to; i = t.find(k, i + 1)) {
    if (i > 0 && !isspace((unsigned char)t[i - 1])) continue;  // ("%YAML:1.0" holds an "L:")
    v = strtol(t.c_str() + i + k.size(), nullptr, 10);
    return true;
  }
  return false;
}

int load_dbow3_yaml(const char *path, const std::string &t, std::vector<VocNode> &nodes, int &k, int &L) {
  const size_t p_nodes = t.find("nodes:"), p_words = t.find("words:");
  long lk = 0, lL = 0;
  if (p_nodes == std::string::npos || p_words == std::string::npos || p_words < p_nodes || !yaml_scalar(t, 0, p_nodes, "k", lk) ||
      !yaml_scalar(t, 0, p_nodes, "L", lL) || lk < 1 || lk > 64 || lL < 1 || lL > 16) {
    vo::set_error("%s: not a DBoW3 cv::FileStorage vocabulary (k / L / nodes / words missing or implausible)", path);
    return VO_ERR_INVALID;
  }
  k = (int)lk, L = (int)lL;
  std::vector<YamlMap> recs, words;
  if (!yaml_flow_maps(t, p_nodes, p_words, recs) || !yaml_flow_maps(t, p_words, t.size(), words) || recs.empty()) {
    vo::set_error("%s: malformed node / word records", path);
    return VO_ERR_INVALID;
  }
  const size_t nn = recs.size() + 1;  // + the root, which is not written
  nodes.assign(nn, VocNode());
  memset(nodes[0].desc, 0, 32);
  std::vector<uint8_t> seen(nn, 0);
  seen[0] = 1;
  for (size_t r = 0; r < recs.size(); r++) {
    const std::string *sid = recs[r].get("nodeId"), *spid = recs[r].get("parentId"), *sw = recs[r].get("weight"),
                      *sd = recs[r].get("descriptor");
    if (!sid || !spid || !sw || !sd) {
      vo::set_error("%s: node record %zu lacks nodeId / parentId / weight / descriptor", path, r);
      return VO_ERR_INVALID;
    }
    const long id = strtol(sid->c_str(), nullptr, 10), pid = strtol(spid->c_str(), nullptr, 10);
    // a parent must be defined before its children (DBoW3's writers emit parent first): a later record for the parent would
    // overwrite the children collected so far, and a node may never be its own ancestor
    if (id <= 0 || (size_t)id >= nn || pid < 0 || (size_t)pid >= nn || seen[id] || pid == id || !seen[pid]) {
      vo::set_error("%s: record %zu names node %ld / parent %ld (of %zu nodes%s)", path, r, id, pid, nn,
                    id > 0 && (size_t)id < nn && seen[id] ? ", twice"
                    : (pid >= 0 && (size_t)pid < nn && !seen[pid]) ? "; the parent is not defined before its child" : "");
      return VO_ERR_INVALID;
    }
    VocNode n;
    n.id = (uint32_t)id, n.parent = (uint32_t)pid, n.weight = strtod(sw->c_str(), nullptr);
    memset(n.desc, 0, 32);
    std::istringstream ds(*sd);
    std::string first;
    ds >> first;
    int cols = 32;
    if (first == "dbw3") {
      int type = -1;
      ds >> type >> cols;
      if (!ds || (type & 7) != 0 || cols != 32) {  // CV_8U, 32 columns: ORB
        vo::set_error("%s: node %ld has a descriptor of type %d with %d columns (need CV_8U x 32)", path, id, type, cols);
        return VO_ERR_INVALID;
      }
    } else {
      ds.clear();
      ds.str(*sd);  // DBoW2-style: the 32 values alone
    }
    for (int b = 0; b < 32; b++) {
      int v = -1;
      ds >> v;
      if (!ds || v < 0 || v > 255) {
        vo::set_error("%s: node %ld: descriptor byte %d missing or out of range", path, id, b);
        return VO_ERR_INVALID;
      }
      n.desc[b] = (uint8_t)v;
    }
    seen[id] = 1;
    nodes[id] = n;
    nodes[pid].children.push_back((uint32_t)id);
  }
  for (size_t w = 0; w < words.size(); w++) {
    const std::string *swid = words[w].get("wordId"), *snid = words[w].get("nodeId");
    const long wid = swid ? strtol(swid->c_str(), nullptr, 10) : -1, nid = snid ? strtol(snid->c_str(), nullptr, 10) : -1;
    if (wid < 0 || (size_t)wid >= words.size() || nid <= 0 || (size_t)nid >= nn) {
      vo::set_error("%s: malformed word table (entry %zu: node %ld, word %ld)", path, w, nid, wid);
      return VO_ERR_INVALID;
    }
    nodes[nid].word_id = (uint32_t)wid;
  }
  return VO_OK;
}

int vocab_to_handle(std::vector<VocNode> &nodes, int L, vo_vocab **out, int *n_nodes, int *n_words) {
  const int N = (int)nodes.size();
  std::vector<int32_t> cs(N + 1, 0), ch, wid(N, -1);
  std::vector<uint8_t> desc((size_t)N * 32, 0);
  std::vector<double> wt(N, 0.0);
  int words = 0;
  for (int i = 0; i < N; i++) {
    cs[i] = (int32_t)ch.size();
    for (uint32_t c : nodes[i].children) ch.push_back((int32_t)c);
    memcpy(&desc[(size_t)i * 32], nodes[i].desc, 32);
    wt[i] = nodes[i].weight;
    if (nodes[i].children.empty() && i > 0) wid[i] = (int32_t)nodes[i].word_id, words++;  // Node::isLeaf()
  }
  cs[N] = (int32_t)ch.size();
  if (n_nodes) *n_nodes = N;
  if (n_words) *n_words = words;
  return vo_vocab_create(out, N, L, cs.data(), ch.data(), desc.data(), wt.data(), wid.data());
}

}  // namespace

extern "C" {

int vo_dataset_open(vo_dataset **out, const char *dataset_dir, int max_frames) {
  if (!out || !dataset_dir) return VO_ERR_INVALID;
  const std::string dir(dataset_dir);
  std::ifstream fin(dir + "/associate.txt");
  if (fin.fail()) {
    vo::set_error("can't find associate file in %s", dataset_dir);  // vo_run.cpp:33-37
    return VO_ERR_INVALID;
  }
  vo_dataset *d = new vo_dataset();
  for (int i = 0; i < max_frames; i++) {  // :43-57, including its quirk: eof is only noticed before a read, so a file that
    if (fin.eof()) break;                 // ends with a newline yields one last record of empty strings
    std::string rt, rf, dt, df;
    fin >> rt >> rf >> dt >> df;
    d->rgb_time.push_back(rt), d->rgb_path.push_back(dir + rf);
    d->depth_time.push_back(dt), d->depth_path.push_back(dir + df);
  }
  *out = d;
  return VO_OK;
}
int vo_dataset_size(const vo_dataset *d) { return d ? (int)d->rgb_time.size() : 0; }
int vo_dataset_entry(const vo_dataset *d, int i, const char **rgb_time, const char **rgb_path, const char **depth_time,
                     const char **depth_path) {
  if (!d || i < 0 || i >= (int)d->rgb_time.size()) return VO_ERR_INVALID;
  if (rgb_time) *rgb_time = d->rgb_time[i].c_str();
  if (rgb_path) *rgb_path = d->rgb_path[i].c_str();
  if (depth_time) *depth_time = d->depth_time[i].c_str();
  if (depth_path) *depth_path = d->depth_path[i].c_str();
  return VO_OK;
}
void vo_dataset_close(vo_dataset *d) { delete d; }

int vo_png_info(const char *path, int *width, int *height, int *channels, int *bit_depth) {
  if (!path) return VO_ERR_INVALID;
  Png P;
  const int rc = png_decode(path, P, true);
  if (rc != VO_OK) return rc;
  if (width) *width = P.w;
  if (height) *height = P.h;
  if (channels) *channels = P.color == 3 ? 3 : P.channels;
  if (bit_depth) *bit_depth = P.depth < 8 ? 8 : P.depth;
  return VO_OK;
}

int vo_png_read(const char *path, int as_bgr, void *dst, size_t dst_bytes) {
  if (!path || !dst) return VO_ERR_INVALID;
  Png P;
  const int rc = png_decode(path, P, false);
  if (rc != VO_OK) return rc;
  const size_t npx = (size_t)P.w * P.h;
  if (P.depth == 16) {  // cv::imread(path, -1): samples as stored, host byte order
    if (dst_bytes < npx * P.channels * 2) return VO_ERR_CAPACITY;
    uint16_t *o = static_cast<uint16_t *>(dst);
    for (size_t i = 0; i < npx * P.channels; i++) o[i] = (uint16_t)((P.pixels[2 * i] << 8) | P.pixels[2 * i + 1]);
    return VO_OK;
  }
  uint8_t *o = static_cast<uint8_t *>(dst);
  if (P.color == 3) {  // palette -> colour
    if (dst_bytes < npx * 3) return VO_ERR_CAPACITY;
    for (size_t i = 0; i < npx; i++) {
      const uint8_t *c = &P.palette[3 * std::min<size_t>(P.pixels[i], P.palette.size() / 3 - 1)];
      o[3 * i] = as_bgr ? c[2] : c[0], o[3 * i + 1] = c[1], o[3 * i + 2] = as_bgr ? c[0] : c[2];
    }
    return VO_OK;
  }
  if (dst_bytes < npx * P.channels) return VO_ERR_CAPACITY;
  if (P.channels >= 3 && as_bgr) {  // cv::imread(path, 1) hands out B, G, R
    for (size_t i = 0; i < npx; i++) {
      const uint8_t *c = &P.pixels[i * P.channels];
      o[i * P.channels] = c[2], o[i * P.channels + 1] = c[1], o[i * P.channels + 2] = c[0];
      if (P.channels == 4) o[i * 4 + 3] = c[3];
    }
  } else {
    memcpy(o, P.pixels.data(), npx * P.channels);
  }
  return VO_OK;
}

int vo_trajectory_write(const char *path, int n, const char *const *timestamps, const double *Twc7) {
  if (!path || n < 0 || (n > 0 && (!timestamps || !Twc7))) return VO_ERR_INVALID;
  std::ofstream f(path);
  if (!f) {
    vo::set_error("cannot write %s", path);
    return VO_ERR_INVALID;
  }
  for (int i = 0; i < n; i++)  // :171-172 / :227-228: translation, then quaternion coefficients x y z w
    f << timestamps[i] << " " << eigen_row(Twc7 + 7 * i, 3) << " " << eigen_row(Twc7 + 7 * i + 3, 4) << std::endl;
  return VO_OK;
}

int vo_tracking_time_stats(const double *seconds, int n_tracked, double *median, double *mean) {
  if (n_tracked < 1 || !seconds || !median || !mean) return VO_ERR_INVALID;
  std::vector<double> t(seconds, seconds + n_tracked);
  std::sort(t.begin(), t.end());  // :138-151
  double total = 0;
  for (double v : t) total += v;
  *median = t[n_tracked / 2];
  *mean = total / n_tracked;
  return VO_OK;
}

}  // extern "C"

namespace {
// ---- QuickLZ 1.5.0, compression level 1, no streaming buffer: the coder DBoW3's Vocabulary::toStream runs over 10000-byte
// chunks of its stream when `compressed` is set (what `vocab.save(path)` does by default, reference src/map.cpp:94).
// DBoW3 and QuickLZ are not vendored under the reference; this restates the published decoder (quicklz.c,
// qlz_decompress_core): a chunk is a 3- or 9-byte header (flags: bit 0 compressed, bit 1 long header, bits 2-3 level;
// then compressed and decompressed size) and a sequence of 32-bit control words, each followed by the items its bits
// announce from the low bit up -- 0: a literal byte, 1: a match.  A level-1 match names no offset: its 12-bit field is the
// hash of the three bytes it starts with, and the source is the most recent earlier position with that hash, so the
// decoder keeps the coder's hash table (positions are hashed as their three bytes become available; the inside of a
// match is not hashed).  Every read and write is bounds-checked: a stream that does not follow the format is refused.
// UNPINNED: no file written by DBoW3 itself is available in this environment -- the tests round-trip through a coder
// restated the same way (tests/qlz_ref.py).
struct QlzReader {
  const uint8_t *p;
  size_t n;
  uint32_t rd(size_t at, int bytes) const {  // little-endian, zero beyond the end (the reference over-reads its padding)
    uint32_t v = 0;
    for (int i = 0; i < bytes; i++)
      if (at + i < n) v |= (uint32_t)p[at + i] << (8 * i);
    return v;
  }
};

bool qlz_decompress_chunk(const uint8_t *src, size_t src_len, std::vector<uint8_t> &out, std::string &why) {
  if (src_len < 3) return why = "chunk shorter than its header", false;
  const unsigned flags = src[0];
  const size_t hdr = (flags & 2) ? 9 : 3;
  if (src_len < hdr) return why = "chunk shorter than its header", false;
  QlzReader R{src, src_len};
  const size_t csize = (flags & 2) ? R.rd(1, 4) : src[1], dsize = (flags & 2) ? R.rd(5, 4) : src[2];
  if (csize != src_len) return why = "compressed size field does not match the chunk", false;
  if (dsize > (64u << 20)) return why = "implausible decompressed size", false;
  out.assign(dsize, 0);
  if (!(flags & 1)) {  // stored
    if (src_len - hdr < dsize) return why = "stored chunk is truncated", false;
    memcpy(out.data(), src + hdr, dsize);
    return true;
  }
  const int level = (flags >> 2) & 3;
  if (level != 1) {
    why = "QuickLZ level " + std::to_string(level) + " stream (only level 1, DBoW3's build default, is decoded)";
    return false;
  }
  if (dsize == 0) return true;
  std::vector<int64_t> table(4096, -1);
  auto hash3 = [&](int64_t at) {
    const uint32_t i = (uint32_t)out[at] | ((uint32_t)out[at + 1] << 8) | ((uint32_t)out[at + 2] << 16);
    return ((i >> 12) ^ i) & 4095u;
  };
  const int64_t last = (int64_t)dsize - 1, last_matchstart = last - 6 - 4;
  int64_t dst = 0, last_hashed = -1;
  size_t sp = hdr;
  uint32_t cword = 1;
  auto hash_upto = [&](int64_t max) {  // positions (last_hashed, max]; position q needs bytes q .. q + 2
    while (last_hashed < max) {
      last_hashed++;
      if (last_hashed + 2 <= last) table[hash3(last_hashed)] = last_hashed;
    }
  };
  static const unsigned bitlut[16] = {4, 0, 1, 0, 2, 0, 1, 0, 3, 0, 1, 0, 2, 0, 1, 0};
  for (;;) {
    if (cword == 1) {
      if (sp + 4 > src_len) return why = "control word beyond the chunk", false;
      cword = R.rd(sp, 4), sp += 4;
      if (cword == 0) return why = "control word without its end marker", false;
    }
    const uint32_t fetch = R.rd(sp, 4);
    if (cword & 1u) {
      cword >>= 1;
      const unsigned hash = (fetch >> 4) & 0xfffu;
      const int64_t from = table[hash];
      size_t mlen;
      if (fetch & 0xfu) mlen = (fetch & 0xfu) + 2, sp += 2;
      else mlen = (fetch >> 16) & 0xffu, sp += 3;
      if (sp > src_len || from < 0 || from >= dst || mlen < 3 || dst + (int64_t)mlen > (int64_t)dsize)
        return why = "match outside the data decoded so far", false;
      for (size_t i = 0; i < mlen; i++) out[dst + i] = out[from + i];  // forward, byte by byte: source and target may overlap
      dst += mlen;
      hash_upto(dst - (int64_t)mlen);
      last_hashed = dst - 1;
    } else if (dst < last_matchstart) {
      const unsigned nlit = bitlut[cword & 0xfu];
      if (sp + nlit > src_len || dst + nlit > (int64_t)dsize) return why = "literals beyond the chunk", false;
      for (unsigned i = 0; i < nlit; i++) out[dst + i] = src[sp + i];
      cword >>= nlit, dst += nlit, sp += nlit;
      hash_upto(dst - 3);
    } else {
      while (dst <= last) {
        if (cword == 1) sp += 4, cword = 1u << 31;
        if (sp >= src_len) return why = "trailing literals beyond the chunk", false;
        out[dst++] = src[sp++];
        cword >>= 1;
      }
      return true;
    }
    if (dst > last) return true;  // (a stream whose last item is a match ends here)
  }
}
}  // namespace

extern "C" {

int vo_vocab_load(const char *path, vo_vocab **out, int *n_nodes, int *n_words, int *branching_k, int *depth_L) {
  if (!path || !out) return VO_ERR_INVALID;
  std::ifstream f(path, std::ios::binary);
  if (!f) {
    vo::set_error("vocabulary file not exist: %s", path);  // vo_run.cpp:80-84
    return VO_ERR_INVALID;
  }
  uint64_t sig = 0;
  f.read(reinterpret_cast<char *>(&sig), 8);
  std::vector<VocNode> nodes;
  int k = 0, L = 0;
  const bool gz = (sig & 0xffff) == 0x8b1f, yaml = memcmp(&sig, "%YAML", 5) == 0;
  if (f && (gz || yaml)) {
    std::string text;
    if (!read_all_gz(path, text) || text.compare(0, 5, "%YAML") != 0) {
      vo::set_error("%s: %s", path, gz ? "gzip stream that does not hold a cv::FileStorage YAML vocabulary" : "unreadable");
      return VO_ERR_INVALID;
    }
    const int yrc = load_dbow3_yaml(path, text, nodes, k, L);
    if (yrc != VO_OK) return yrc;
  } else if (f && sig == 88877711233ULL) {
    // DBoW3 Vocabulary::toStream (the library is not vendored under the reference; layout restated from its published
    // source): magic, bool compressed, uint32 node count nn; then k, L, scoring, weighting (ints); then nn - 1 records --
    // the root is implicit -- in the writer's depth-first order: node id, parent id (uint32), weight (double), descriptor
    // (int cols, rows, type, then cols x elemSize bytes); then uint32 word count and per word (node id, word id).
    // Children are attached to their parent in FILE order (fromStream does `m_nodes[parent].children.push_back(id)` per
    // record), which is the order Vocabulary::transform visits them in -- ties in the Hamming distance go to the first.
    // Compressed streams (what `Vocabulary::save(path)` writes by default): QuickLZ level-1 chunks, decoded by
    // qlz_decompress_chunk above.
    char compressed = 0;
    uint32_t nn = 0;
    f.read(&compressed, 1);
    f.read(reinterpret_cast<char *>(&nn), 4);
    if (!f || nn == 0 || nn > 50u * 1000 * 1000) {
      vo::set_error("%s: implausible node count %u", path, nn);
      return VO_ERR_INVALID;
    }
    std::istringstream mem;
    std::istream *in = &f;
    if (compressed) {
      // uint32 chunk count, then the chunks back to back; each is read the way fromStream reads it: 9 bytes, from which
      // the chunk's size follows, then the rest (a chunk is never shorter than 9 bytes)
      uint32_t n_chunks = 0;
      f.read(reinterpret_cast<char *>(&n_chunks), 4);
      if (!f || n_chunks == 0 || n_chunks > (1u << 20)) {
        vo::set_error("%s: implausible chunk count %u in a compressed DBoW3 vocabulary", path, n_chunks);
        return VO_ERR_INVALID;
      }
      std::string body;
      std::vector<uint8_t> chunk, plain;
      for (uint32_t c = 0; c < n_chunks; c++) {
        // a chunk's header is 3 bytes (flags, compressed size, decompressed size: inputs below 216 bytes) or 9 (32-bit sizes,
        // flag bit 1); the last chunk of a stream can be as short as 4 bytes, so only what the header form needs is read
        // before the size is known (reading 9 bytes up front refused such files)
        chunk.assign(9, 0);
        f.read(reinterpret_cast<char *>(chunk.data()), 3);
        const bool long_hdr = (chunk[0] & 2) != 0;
        size_t have = 3;
        if (long_hdr) {
          f.read(reinterpret_cast<char *>(chunk.data() + 3), 6);
          have = 9;
        }
        const size_t csize = long_hdr ? ((size_t)chunk[1] | (size_t)chunk[2] << 8 | (size_t)chunk[3] << 16 | (size_t)chunk[4] << 24) : chunk[1];
        if (!f || csize < have + 1 || csize > (1u << 24)) {
          vo::set_error("%s: chunk %u of the compressed vocabulary has size %zu", path, c, csize);
          return VO_ERR_INVALID;
        }
        chunk.resize(std::max<size_t>(csize, 9));
        f.read(reinterpret_cast<char *>(chunk.data() + have), (std::streamsize)(csize - have));
        std::string why;
        if (!f || !qlz_decompress_chunk(chunk.data(), csize, plain, why)) {
          vo::set_error("%s: chunk %u of the compressed (QuickLZ) vocabulary cannot be decoded: %s -- re-save it with "
                        "Vocabulary::save(path, false)", path, c, f ? why.c_str() : "truncated file");
          return VO_ERR_INVALID;
        }
        body.append(reinterpret_cast<const char *>(plain.data()), plain.size());
      }
      mem.str(body);
      in = &mem;
    }
    int32_t hdr[4];
    in->read(reinterpret_cast<char *>(hdr), 16);
    k = hdr[0], L = hdr[1];
    if (!*in || k < 1 || k > 64 || L < 1 || L > 16) {
      vo::set_error("%s: implausible branching factor / depth %d / %d", path, k, L);
      return VO_ERR_INVALID;
    }
    nodes.resize(nn);
    nodes[0].id = 0, nodes[0].parent = 0, nodes[0].weight = 0;
    memset(nodes[0].desc, 0, 32);
    std::vector<uint8_t> seen(nn, 0);
    seen[0] = 1;
    for (uint32_t i = 1; i < nn; i++) {
      VocNode n;
      int32_t cols = 0, rows = 0, type = 0;
      in->read(reinterpret_cast<char *>(&n.id), 4);
      in->read(reinterpret_cast<char *>(&n.parent), 4);
      in->read(reinterpret_cast<char *>(&n.weight), 8);
      in->read(reinterpret_cast<char *>(&cols), 4);
      in->read(reinterpret_cast<char *>(&rows), 4);
      in->read(reinterpret_cast<char *>(&type), 4);
      if (!*in) break;
      if (n.id == 0 || n.id >= nn || n.parent >= nn || seen[n.id] || n.parent == n.id || !seen[n.parent]) {
        vo::set_error("%s: record %u names node %u / parent %u (of %u nodes%s)", path, i, n.id, n.parent, nn,
                      n.id < nn && seen[n.id] ? ", twice"
                      : (n.parent < nn && !seen[n.parent]) ? "; the parent is not defined before its child" : "");
        return VO_ERR_INVALID;
      }
      memset(n.desc, 0, 32);
      if ((type & 7) != 0 || cols != 32 || rows != 1) {  // CV_8U, 1 x 32: ORB
        vo::set_error("%s: node %u has a %d x %d descriptor of type %d (need 1 x 32 CV_8U)", path, n.id, rows, cols, type);
        return VO_ERR_INVALID;
      }
      in->read(reinterpret_cast<char *>(n.desc), 32);
      seen[n.id] = 1;
      n.children.clear();
      nodes[n.id] = n;
      nodes[n.parent].children.push_back(n.id);
    }
    uint32_t nw = 0;
    in->read(reinterpret_cast<char *>(&nw), 4);
    if (!*in || nw > nn) {
      vo::set_error("%s: truncated vocabulary", path);
      return VO_ERR_INVALID;
    }
    for (uint32_t i = 0; i < nw; i++) {
      uint32_t nid = 0, wid = 0;
      in->read(reinterpret_cast<char *>(&nid), 4);
      in->read(reinterpret_cast<char *>(&wid), 4);
      if (!*in || nid >= nn || wid >= nw) {
        vo::set_error("%s: malformed word table (entry %u: node %u, word %u)", path, i, nid, wid);
        return VO_ERR_INVALID;
      }
      nodes[nid].word_id = wid;
    }
  } else {
    // ORB-SLAM2 text vocabulary: "k L scoring weighting", then one line per node: parent is_leaf 32 bytes weight
    f.clear();
    f.seekg(0);
    int scoring = 0, weighting = 0;
    std::string line;
    if (!std::getline(f, line)) return VO_ERR_INVALID;
    std::istringstream h(line);
    if (!(h >> k >> L >> scoring >> weighting) || k < 1 || k > 20 || L < 1 || L > 10) {
      vo::set_error("%s: neither a DBoW3 binary nor an ORB-SLAM2 text vocabulary", path);
      return VO_ERR_INVALID;
    }
    nodes.resize(1);
    uint32_t words = 0;
    while (std::getline(f, line)) {
      if (line.empty()) continue;
      std::istringstream s(line);
      VocNode n;
      int parent = 0, leaf = 0;
      s >> parent >> leaf;
      for (int b = 0; b < 32; b++) {
        int v = 0;
        s >> v;
        n.desc[b] = (uint8_t)v;
      }
      s >> n.weight;
      if (!s || parent < 0 || parent >= (int)nodes.size()) {
        vo::set_error("%s: malformed node line %zu", path, nodes.size());
        return VO_ERR_INVALID;
      }
      n.id = (uint32_t)nodes.size(), n.parent = (uint32_t)parent;
      if (leaf) n.word_id = words++;
      nodes[parent].children.push_back(n.id);
      nodes.push_back(n);
    }
  }
  if (nodes.empty()) return VO_ERR_INVALID;
  if (branching_k) *branching_k = k;
  if (depth_L) *depth_L = L;
  return vocab_to_handle(nodes, L, out, n_nodes, n_words);
}

}  // extern "C"
