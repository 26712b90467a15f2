// spanfile_pager.cpp -- implementation of include/syzgy_pager.h: read a SyzgyDB
// collection file into rows + ids.  Host code only; the reference lines each
// step restates are cited in the header.
#include "../../include/syzgy_pager.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr uint32_t kActiveMagic = 0x5350414E;  // 'SPAN' (spanfile.go:58-61)
constexpr uint32_t kFreeMagic = 0x46524545;    // 'FREE'
constexpr uint64_t kMinSpanLength = 15;        // spanfile.go:63

uint32_t g_crc_table[8][256];
std::atomic<bool> g_crc_ready{false};

void crc_init()
{
    if (g_crc_ready.load()) return;
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
        g_crc_table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++)
            g_crc_table[t][i] = (g_crc_table[t - 1][i] >> 8) ^ g_crc_table[0][g_crc_table[t - 1][i] & 0xFF];
    g_crc_ready.store(true);
}

// CRC32-IEEE (hash/crc32.ChecksumIEEE, spanfile.go:836-838), slicing-by-8
uint32_t crc32_ieee(const uint8_t *p, size_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t a, b;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        a ^= c;
        c = g_crc_table[7][a & 0xFF] ^ g_crc_table[6][(a >> 8) & 0xFF] ^ g_crc_table[5][(a >> 16) & 0xFF] ^
            g_crc_table[4][a >> 24] ^ g_crc_table[3][b & 0xFF] ^ g_crc_table[2][(b >> 8) & 0xFF] ^
            g_crc_table[1][(b >> 16) & 0xFF] ^ g_crc_table[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = g_crc_table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

uint32_t be32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

// read7Code, spanfile.go:627-636
bool read7(const uint8_t *buf, size_t len, size_t *at, uint64_t *out)
{
    uint64_t r = 0;
    for (size_t o = *at; o < len; o++) {
        const uint64_t d = buf[o];
        r = (r << 7) | (d & 0x7F);
        if ((d & 0x80) == 0) {
            *at = o + 1;
            *out = r;
            return true;
        }
    }
    return false;
}

struct Record {
    uint32_t seq = 0;
    const uint8_t *meta = nullptr;
    uint64_t meta_len = 0;
    const uint8_t *vec = nullptr;
    uint64_t vec_len = 0;
    bool has_vec = false;
};

struct ParsedSpan {
    bool ok = false;
    std::string id;
    Record rec;
};

// parseSpan, spanfile.go:730-818 (checksum verified by the caller)
ParsedSpan parse_span(const uint8_t *data, size_t len)
{
    ParsedSpan ps;
    size_t at = 8;
    uint64_t seq, idlen;
    if (!read7(data, len, &at, &seq)) return ps;
    if (!read7(data, len, &at, &idlen)) return ps;
    if (at + idlen > len) return ps;
    ps.id.assign(reinterpret_cast<const char *>(data + at), (size_t)idlen);
    at += (size_t)idlen;
    if (at >= len) return ps;
    const int n_streams = data[at++];
    ps.rec.seq = (uint32_t)seq;
    for (int i = 0; i < n_streams; i++) {
        if (at >= len) return ps;
        const uint8_t sid = data[at++];
        uint64_t sl;
        if (!read7(data, len, &at, &sl)) return ps;
        if (at + sl > len) return ps;
        // streams are addressed by POSITION in the reference (collection.go:476-477)
        if (i == 0) {
            ps.rec.meta = data + at;
            ps.rec.meta_len = sl;
        } else if (i == 1) {
            ps.rec.vec = data + at;
            ps.rec.vec_len = sl;
            ps.rec.has_vec = true;
        }
        (void)sid;
        at += (size_t)sl;
    }
    if (at + 4 > len) return ps;
    ps.ok = true;
    return ps;
}

bool json_int(const std::string &js, const char *key, long *out)
{
    const std::string k = std::string("\"") + key + "\"";
    size_t p = js.find(k);
    if (p == std::string::npos) return false;
    p = js.find(':', p + k.size());
    if (p == std::string::npos) return false;
    p++;
    while (p < js.size() && (js[p] == ' ' || js[p] == '\t')) p++;
    char *end = nullptr;
    const long v = strtol(js.c_str() + p, &end, 10);
    if (end == js.c_str() + p) return false;
    *out = v;
    return true;
}

int64_t row_bytes_of(int bits, int dim)
{
    switch (bits) {
    case 4: return ((int64_t)dim + 1) / 2;
    case 8: return dim;
    case 16: return (int64_t)dim * 2;
    case 32: return (int64_t)dim * 4;
    case 64: return (int64_t)dim * 8;
    default: return -1;
    }
}

}  // namespace

struct szg_pager {
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t size = 0;
    int dim = 0, bits = 0, metric = 0;
    int64_t row_bytes = 0;
    uint64_t skipped = 0;
    std::vector<uint64_t> ids;    // visit order
    std::vector<Record> recs;     // same order
};

extern "C" {

void szg_pager_close(szg_pager *p)
{
    if (!p) return;
    if (p->map && p->size) munmap(const_cast<uint8_t *>(p->map), p->size);
    if (p->fd >= 0) close(p->fd);
    delete p;
}

int szg_pager_open(szg_pager **out, const char *path, int n_threads)
{
    if (!out || !path) return SZG_E_INVALID;
    *out = nullptr;
    crc_init();
    szg_pager *p = new szg_pager();
    p->fd = open(path, O_RDONLY);
    if (p->fd < 0) {
        szg_pager_close(p);
        return SZG_E_IO;
    }
    struct stat st;
    if (fstat(p->fd, &st) != 0 || st.st_size <= 0) {
        szg_pager_close(p);
        return SZG_E_IO;
    }
    p->size = (size_t)st.st_size;
    void *m = mmap(nullptr, p->size, PROT_READ, MAP_PRIVATE, p->fd, 0);
    if (m == MAP_FAILED) {
        p->map = nullptr;
        szg_pager_close(p);
        return SZG_E_IO;
    }
    p->map = static_cast<const uint8_t *>(m);

    // pass 1 (sequential, headers only): span boundaries, spanfile.go:282-357
    struct Span {
        size_t off;
        uint32_t len;
    };
    std::vector<Span> spans;
    size_t off = 0;
    while (off < p->size) {
        if (off + kMinSpanLength > p->size) break;
        const uint32_t magic = be32(p->map + off);
        if (magic == 0) break;  // rest of the file is free space
        const uint32_t len = be32(p->map + off + 4);
        if ((uint64_t)off + len > p->size) break;
        if (len == 0) break;    // "length is 0; can't continue"
        if (magic == kActiveMagic) spans.push_back(Span{off, len});
        off += len;             // FREE and unknown magics are skipped by their length
    }

    // pass 2 (parallel): checksum + parse
    std::vector<ParsedSpan> parsed(spans.size());
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(nt, 64));
    if (spans.size() < 1024) nt = 1;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t i0 = next.fetch_add(256);
            if (i0 >= spans.size()) break;
            const size_t i1 = std::min(spans.size(), i0 + 256);
            for (size_t i = i0; i < i1; i++) {
                const uint8_t *d = p->map + spans[i].off;
                const uint32_t len = spans[i].len;
                if (len < kMinSpanLength) continue;
                if (crc32_ieee(d, len - 4) != be32(d + len - 4)) continue;  // verifyChecksum
                parsed[i] = parse_span(d, len);
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();

    // pass 3 (sequential): highest sequence number per record id wins (spanfile.go:337-341)
    std::map<std::string, Record> index;
    for (size_t i = 0; i < parsed.size(); i++) {
        if (!parsed[i].ok) {
            p->skipped++;
            continue;
        }
        auto it = index.find(parsed[i].id);
        if (it == index.end() || parsed[i].rec.seq > it->second.seq) index[parsed[i].id] = parsed[i].rec;
    }

    // header record "" : CollectionOptions JSON in stream 0 (collection.go:241-252)
    auto h = index.find("");
    if (h == index.end() || !h->second.meta) {
        szg_pager_close(p);
        return SZG_E_FORMAT;
    }
    const std::string js(reinterpret_cast<const char *>(h->second.meta), (size_t)h->second.meta_len);
    long v;
    if (!json_int(js, "dimension_count", &v)) {
        szg_pager_close(p);
        return SZG_E_FORMAT;
    }
    p->dim = (int)v;
    p->bits = json_int(js, "quantization", &v) ? (int)v : 64;
    p->metric = json_int(js, "distance_method", &v) ? (int)v : 0;
    p->row_bytes = row_bytes_of(p->bits, p->dim);
    if (p->dim <= 0 || p->row_bytes <= 0) {
        szg_pager_close(p);
        return SZG_E_FORMAT;
    }

    // std::map iterates its string keys in sort.Strings order (byte-wise), which is
    // IterateSortedRecords' order (spanfile.go:540-560)
    for (const auto &kv : index) {
        if (kv.first.empty()) continue;
        char *end = nullptr;
        const unsigned long long id = strtoull(kv.first.c_str(), &end, 10);
        if (*end != '\0' || end == kv.first.c_str()) continue;  // strconv.ParseUint failure: skipped (collection.go:676)
        if (!kv.second.has_vec || (int64_t)kv.second.vec_len < p->row_bytes) continue;
        p->ids.push_back((uint64_t)id);
        p->recs.push_back(kv.second);
    }
    *out = p;
    return SZG_OK;
}

int szg_pager_options(const szg_pager *p, int *dim, int *quant_bits, int *metric)
{
    if (!p) return SZG_E_INVALID;
    if (dim) *dim = p->dim;
    if (quant_bits) *quant_bits = p->bits;
    if (metric) *metric = p->metric;
    return SZG_OK;
}

uint64_t szg_pager_count(const szg_pager *p) { return p ? p->ids.size() : 0; }
uint64_t szg_pager_skipped(const szg_pager *p) { return p ? p->skipped : 0; }

int szg_pager_ids(const szg_pager *p, uint64_t *ids)
{
    if (!p || (!ids && !p->ids.empty())) return SZG_E_INVALID;
    if (!p->ids.empty()) memcpy(ids, p->ids.data(), p->ids.size() * sizeof(uint64_t));
    return SZG_OK;
}

int szg_pager_vectors(const szg_pager *p, uint8_t *out, uint64_t capacity_bytes)
{
    if (!p) return SZG_E_INVALID;
    const uint64_t need = (uint64_t)p->ids.size() * (uint64_t)p->row_bytes;
    if (capacity_bytes < need || (!out && need)) return SZG_E_INVALID;
    for (size_t r = 0; r < p->recs.size(); r++)
        memcpy(out + r * (size_t)p->row_bytes, p->recs[r].vec, (size_t)p->row_bytes);
    return SZG_OK;
}

int szg_pager_metadata(const szg_pager *p, uint64_t row, const uint8_t **data, uint64_t *len)
{
    if (!p || !data || !len) return SZG_E_INVALID;
    if (row >= p->recs.size()) return SZG_E_RANGE;
    *data = p->recs[row].meta;
    *len = p->recs[row].meta_len;
    return SZG_OK;
}

int szg_pager_load(const szg_pager *p, szg_index *ix)
{
    if (!p || !ix) return SZG_E_INVALID;
    std::vector<uint8_t> buf((size_t)p->ids.size() * (size_t)p->row_bytes);
    int rc = szg_pager_vectors(p, buf.data(), buf.size());
    if (rc) return rc;
    return szg_index_load(ix, buf.data(), p->ids.size());
}

}  // extern "C"
