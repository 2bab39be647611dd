// parquet.cpp — host side of the Parquet column-chunk -> device loader: footer / page-header parsing
// (Thrift compact protocol, restated from the Parquet format specification; the reference reads the same files through
// github.com/xitongsys/parquet-go v1.6.2, bowparquet.go:44-153), page table, upload of the column's bytes as they lie in
// the file, launch of the decode kernels (parquet_decode.hip).
//
// Scope: what the reference writes (bowparquet.go:326-338: SNAPPY, PLAIN, data page v1, RLE definition levels, flat schema
// of OPTIONAL columns) and what pyarrow / pandas write by default (a dictionary page per column chunk, RLE_DICTIONARY data
// pages, PLAIN fall-back pages, data page v2 when asked for); INT64 and DOUBLE columns (the device path's types).  Other codecs /
// encodings and nested schemas are declined with BOWGPU_ERR_UNSUPPORTED.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <exception>

#include "common.h"

namespace bowgpu {

struct PqPage {  // must match parquet_decode.hip
    int64_t src_off, raw_off, row0;
    int32_t comp_size, raw_size, num_values, compressed;
    int32_t kind, dict_count;
    int64_t dict_off, idx_off;
    int64_t lv_off;
    int32_t lv_len, v2;
    int32_t dict_in_raw, _pad;
};
int launch_parquet_decode(Ctx *c, const uint8_t *chunk, const PqPage *pages, int64_t npages, bool any_compressed, uint8_t *raw,
                          int optional, bool any_dict, uint32_t *indices, uint64_t *out_values, uint32_t *out_valid,
                          unsigned long long *valid_count, uint32_t *status);

namespace {

// ---------------------------------------------------------------- Thrift compact protocol (read side)
struct TReader {
    const uint8_t *b;
    size_t n, p = 0;
    bool ok = true;
    int depth = 0;   // nesting of skip() / skip_struct(): a corrupt footer must not overflow the stack
    uint8_t byte() { if (p >= n) { ok = false; return 0; } return b[p++]; }
    uint64_t varint() {
        uint64_t r = 0;
        int sh = 0;
        while (ok) {
            const uint8_t c = byte();
            r |= (uint64_t)(c & 0x7f) << sh;
            if (!(c & 0x80)) break;
            sh += 7;
            if (sh > 63) { ok = false; break; }
        }
        return r;
    }
    int64_t zigzag() { const uint64_t v = varint(); return (int64_t)(v >> 1) ^ -(int64_t)(v & 1); }
    std::string binary() {
        const uint64_t len = varint();
        if (!ok || len > n - p) { ok = false; return std::string(); }   // (p <= n always; no wrap for a 64-bit len)
        std::string s(reinterpret_cast<const char *>(b + p), (size_t)len);
        p += (size_t)len;
        return s;
    }
    void skip(int type) {
        if (!ok) return;
        if (++depth > 64) { ok = false; --depth; return; }
        skip1(type);
        --depth;
    }
    void skip1(int type) {
        switch (type) {
        case 1: case 2: break;                         // bool in the field header
        case 3: byte(); break;
        case 4: case 5: case 6: varint(); break;
        case 7: p += 8; if (p > n) ok = false; break;
        case 8: binary(); break;
        case 9: case 10: {
            const uint8_t h = byte();
            uint64_t cnt = h >> 4;
            if (cnt == 15) cnt = varint();
            if (cnt > n - p) { ok = false; break; }   // every element takes at least one byte
            for (uint64_t i = 0; i < cnt && ok; i++) { if ((h & 15) <= 2) byte(); else skip(h & 15); }  // (list<bool>: one byte each)
            break;
        }
        case 11: {
            const uint64_t cnt = varint();
            if (cnt > n - p) { ok = false; break; }   // (bool keys / values take no bytes: the count itself is bounded by the input)
            if (cnt) { const uint8_t kv = byte(); for (uint64_t i = 0; i < cnt && ok; i++) { skip(kv >> 4); skip(kv & 15); } }
            break;
        }
        case 12: skip_struct(); break;
        default: ok = false;
        }
    }
    // iterate the fields of a struct: returns false at the stop byte
    bool field(int16_t *fid, int *type) {
        const uint8_t h = byte();
        if (!ok || h == 0) return false;
        const int delta = h >> 4;
        *type = h & 15;
        *fid = delta ? (int16_t)(*fid + delta) : (int16_t)zigzag();
        return ok;
    }
    void skip_struct() { int16_t fid = 0; int t; while (ok && field(&fid, &t)) skip(t); }
    // list header: element type and count
    uint64_t list(int *etype) { const uint8_t h = byte(); uint64_t cnt = h >> 4; if (cnt == 15) cnt = varint(); *etype = h & 15; return cnt; }
};

struct PqColumnChunk {
    int32_t type = -1, codec = 0;
    int64_t num_values = 0, total_compressed = 0, data_page_offset = 0, dictionary_page_offset = -1;
    std::vector<int32_t> encodings;
};
struct PqRowGroup { std::vector<PqColumnChunk> cols; int64_t num_rows = 0; };
struct PqSchemaCol { std::string name; int32_t type = -1, repetition = 0; };

}  // namespace

struct ParquetFile {
    std::string path;
    int fd = -1;
    const uint8_t *map = nullptr;       // the whole file, mapped read-only: page headers are parsed in place and the column
    int64_t size = 0, num_rows = 0;     // chunks go to the device straight from the page cache
    std::vector<PqSchemaCol> cols;      // leaf columns of a flat schema, in order
    std::vector<PqRowGroup> groups;
    bool flat = true;
    ~ParquetFile() {
        if (map) munmap(const_cast<uint8_t *>(map), (size_t)size);
        if (fd >= 0) close(fd);
    }
};

namespace {

int parse_footer(ParquetFile *pf, const uint8_t *buf, size_t len) {
    TReader r{buf, len};
    int16_t fid = 0;
    int t;
    while (r.field(&fid, &t)) {
        if (fid == 2 && t == 9) {  // schema: list<SchemaElement>; element 0 is the root
            int et;
            const uint64_t cnt = r.list(&et);
            for (uint64_t i = 0; i < cnt && r.ok; i++) {
                PqSchemaCol sc;
                int32_t num_children = 0;
                int16_t f2 = 0;
                int t2;
                while (r.field(&f2, &t2)) {
                    if (f2 == 1) sc.type = (int32_t)r.zigzag();
                    else if (f2 == 3) sc.repetition = (int32_t)r.zigzag();
                    else if (f2 == 4) sc.name = r.binary();
                    else if (f2 == 5) num_children = (int32_t)r.zigzag();
                    else r.skip(t2);
                }
                if (i == 0) continue;                      // root
                if (num_children > 0 || sc.repetition == 2) pf->flat = false;  // nested / REPEATED: outside the loader
                pf->cols.push_back(sc);
            }
        } else if (fid == 3) {
            pf->num_rows = r.zigzag();
        } else if (fid == 4 && t == 9) {  // row_groups
            int et;
            const uint64_t cnt = r.list(&et);
            for (uint64_t g = 0; g < cnt && r.ok; g++) {
                PqRowGroup rg;
                int16_t f2 = 0;
                int t2;
                while (r.field(&f2, &t2)) {
                    if (f2 == 1 && t2 == 9) {
                        int et2;
                        const uint64_t nc = r.list(&et2);
                        for (uint64_t c = 0; c < nc && r.ok; c++) {
                            PqColumnChunk cc;
                            int16_t f3 = 0;
                            int t3;
                            while (r.field(&f3, &t3)) {
                                if (f3 == 3 && t3 == 12) {  // ColumnMetaData
                                    int16_t f4 = 0;
                                    int t4;
                                    while (r.field(&f4, &t4)) {
                                        if (f4 == 1) cc.type = (int32_t)r.zigzag();
                                        else if (f4 == 2 && t4 == 9) { int e; const uint64_t ne = r.list(&e); for (uint64_t k = 0; k < ne && r.ok; k++) cc.encodings.push_back((int32_t)r.zigzag()); }
                                        else if (f4 == 4) cc.codec = (int32_t)r.zigzag();
                                        else if (f4 == 5) cc.num_values = r.zigzag();
                                        else if (f4 == 7) cc.total_compressed = r.zigzag();
                                        else if (f4 == 9) cc.data_page_offset = r.zigzag();
                                        else if (f4 == 11) cc.dictionary_page_offset = r.zigzag();
                                        else r.skip(t4);
                                    }
                                } else {
                                    r.skip(t3);
                                }
                            }
                            rg.cols.push_back(cc);
                        }
                    } else if (f2 == 3) {
                        rg.num_rows = r.zigzag();
                    } else {
                        r.skip(t2);
                    }
                }
                pf->groups.push_back(rg);
            }
        } else {
            r.skip(t);
        }
    }
    if (!r.ok) return fail(BOWGPU_ERR_ARG, "parquet: malformed footer in '%s'", pf->path.c_str());
    return 0;
}

struct PageHdr {
    int32_t type = -1, raw_size = 0, comp_size = 0, num_values = 0, encoding = -1, def_encoding = -1, dict_values = 0, dict_encoding = -1;
    int32_t def_len = 0, rep_len = 0, v2_compressed = 1;  // data page v2
    size_t hdr_len = 0;
    bool v2 = false;
};

bool parse_page_header(const uint8_t *b, size_t n, PageHdr *h) {
    TReader r{b, n};
    int16_t fid = 0;
    int t;
    while (r.field(&fid, &t)) {
        if (fid == 1) h->type = (int32_t)r.zigzag();
        else if (fid == 2) h->raw_size = (int32_t)r.zigzag();
        else if (fid == 3) h->comp_size = (int32_t)r.zigzag();
        else if (fid == 5 && t == 12) {
            int16_t f2 = 0;
            int t2;
            while (r.field(&f2, &t2)) {
                if (f2 == 1) h->num_values = (int32_t)r.zigzag();
                else if (f2 == 2) h->encoding = (int32_t)r.zigzag();
                else if (f2 == 3) h->def_encoding = (int32_t)r.zigzag();
                else r.skip(t2);
            }
        } else if (fid == 7 && t == 12) {  // dictionary_page_header
            int16_t f2 = 0;
            int t2;
            while (r.field(&f2, &t2)) {
                if (f2 == 1) h->dict_values = (int32_t)r.zigzag();
                else if (f2 == 2) h->dict_encoding = (int32_t)r.zigzag();
                else r.skip(t2);
            }
        } else if (fid == 8 && t == 12) {  // data_page_header_v2
            h->v2 = true;
            int16_t f2 = 0;
            int t2;
            while (r.field(&f2, &t2)) {
                if (f2 == 1) h->num_values = (int32_t)r.zigzag();
                else if (f2 == 4) h->encoding = (int32_t)r.zigzag();
                else if (f2 == 5) h->def_len = (int32_t)r.zigzag();
                else if (f2 == 6) h->rep_len = (int32_t)r.zigzag();
                else if (f2 == 7) h->v2_compressed = t2 == 1 ? 1 : 0;  // bool lives in the field header
                else r.skip(t2);
            }
        }
        else r.skip(t);
    }
    h->hdr_len = r.p;
    return r.ok;
}

}  // namespace
}  // namespace bowgpu

using namespace bowgpu;

extern "C" {

// (no C++ exception may cross the C ABI: a corrupt footer can still make a container throw length_error / bad_alloc)
static int parquet_open_impl(const char *path, bowgpu_parquet **handle);
static int parquet_read_column_impl(bowgpu_parquet *handle, int32_t i, bowgpu_out *out);

int bowgpu_parquet_open(const char *path, bowgpu_parquet **handle) {
    if (!path || !handle) return fail(BOWGPU_ERR_ARG, "null argument");
    try { return parquet_open_impl(path, handle); }
    catch (const std::exception &e) { *handle = nullptr; return fail(BOWGPU_ERR_ARG, "parquet: malformed file '%s' (%s)", path, e.what()); }
    catch (...) { *handle = nullptr; return fail(BOWGPU_ERR_ARG, "parquet: malformed file '%s'", path); }
}

static int parquet_open_impl(const char *path, bowgpu_parquet **handle) {
    *handle = nullptr;
    struct Guard { ParquetFile *p; ~Guard() { delete p; } } guard{new ParquetFile()};
    ParquetFile *pf = guard.p;
    pf->path = path;
    pf->fd = open(path, O_RDONLY);
    if (pf->fd < 0) return fail(BOWGPU_ERR_ARG, "parquet: cannot open '%s'", path);
    struct stat st;
    int rc = 0;
    if (fstat(pf->fd, &st) != 0 || st.st_size < 12) rc = fail(BOWGPU_ERR_ARG, "parquet: '%s' is too short", path);
    if (!rc) {
        pf->size = (int64_t)st.st_size;
        void *m = mmap(nullptr, (size_t)pf->size, PROT_READ, MAP_PRIVATE, pf->fd, 0);
        if (m == MAP_FAILED) rc = fail(BOWGPU_ERR_ARG, "parquet: cannot map '%s'", path);
        else pf->map = reinterpret_cast<const uint8_t *>(m);
    }
    if (!rc && (memcmp(pf->map + pf->size - 4, "PAR1", 4) != 0 || memcmp(pf->map, "PAR1", 4) != 0)) rc = fail(BOWGPU_ERR_ARG, "parquet: '%s' lacks the PAR1 magic", path);
    if (!rc) {
        const uint8_t *tail = pf->map + pf->size - 8;
        const uint32_t flen = (uint32_t)tail[0] | ((uint32_t)tail[1] << 8) | ((uint32_t)tail[2] << 16) | ((uint32_t)tail[3] << 24);
        if ((int64_t)flen + 12 > pf->size) rc = fail(BOWGPU_ERR_ARG, "parquet: bad footer length in '%s'", path);
        if (!rc) rc = parse_footer(pf, pf->map + pf->size - 8 - flen, flen);
    }
    if (!rc)
        for (const PqRowGroup &g : pf->groups)
            if (g.cols.size() != pf->cols.size()) { rc = fail(BOWGPU_ERR_UNSUPPORTED, "parquet: nested schema in '%s' is outside the loader", path); break; }
    if (rc) return rc;
    guard.p = nullptr;
    *handle = reinterpret_cast<bowgpu_parquet *>(pf);
    return 0;
}

int bowgpu_parquet_close(bowgpu_parquet *handle) {
    delete reinterpret_cast<ParquetFile *>(handle);
    return 0;
}

int bowgpu_parquet_info(const bowgpu_parquet *handle, int64_t *num_rows, int32_t *num_columns) {
    const ParquetFile *pf = reinterpret_cast<const ParquetFile *>(handle);
    if (!pf || !num_rows || !num_columns) return fail(BOWGPU_ERR_ARG, "null argument");
    *num_rows = pf->num_rows;
    *num_columns = (int32_t)pf->cols.size();
    return 0;
}

int bowgpu_parquet_column(const bowgpu_parquet *handle, int32_t i, char *name, int32_t name_cap, int32_t *type, int32_t *optional) {
    const ParquetFile *pf = reinterpret_cast<const ParquetFile *>(handle);
    if (!pf) return fail(BOWGPU_ERR_ARG, "null argument");
    if (i < 0 || i >= (int32_t)pf->cols.size()) return fail(BOWGPU_ERR_BAD_COL, "parquet: no column with index %d", i);
    const PqSchemaCol &sc = pf->cols[i];
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s", sc.name.c_str());
    if (type) *type = sc.type == 2 ? BOWGPU_INT64 : sc.type == 5 ? BOWGPU_FLOAT64 : sc.type == 0 ? BOWGPU_BOOLEAN : sc.type == 6 ? BOWGPU_STRING : -1;
    if (optional) *optional = sc.repetition == 1 ? 1 : 0;
    return 0;
}

int bowgpu_parquet_read_column(bowgpu_parquet *handle, int32_t i, bowgpu_out *out) {
    try { return parquet_read_column_impl(handle, i, out); }
    catch (const std::exception &e) { return fail(BOWGPU_ERR_ARG, "parquet: malformed file (%s)", e.what()); }
    catch (...) { return fail(BOWGPU_ERR_ARG, "parquet: malformed file"); }
}

static int parquet_read_column_impl(bowgpu_parquet *handle, int32_t i, bowgpu_out *out) {
    ParquetFile *pf = reinterpret_cast<ParquetFile *>(handle);
    if (!pf || !out) return fail(BOWGPU_ERR_ARG, "null argument");
    if (i < 0 || i >= (int32_t)pf->cols.size()) return fail(BOWGPU_ERR_BAD_COL, "parquet: no column with index %d", i);
    if (!pf->flat) return fail(BOWGPU_ERR_UNSUPPORTED, "parquet: nested / repeated schema is outside the loader");
    const PqSchemaCol &sc = pf->cols[i];
    if (sc.type != 2 && sc.type != 5) return fail(BOWGPU_ERR_UNSUPPORTED, "parquet: column '%s' is not INT64 / DOUBLE (physical type %d)", sc.name.c_str(), sc.type);
    const int optional = sc.repetition == 1 ? 1 : 0;
    const int64_t n = pf->num_rows;
    Ctx *c;
    BG_TRY(ctx_get(&c));
    DevOut dout;
    BG_TRY(devout_prepare(c, out, n, &dout, 0));
    const int32_t otype = sc.type == 2 ? BOWGPU_INT64 : BOWGPU_FLOAT64;
    if (n == 0) { BG_TRY(devout_finish(c, &dout, 0, otype, 0)); return 0; }

    // ---- page table: walk the page headers of this column in every row group (in place, on the mapped file); the chunks are
    // laid out back to back in the device buffer
    struct Span { int64_t file_off, len, dev_off; };
    std::vector<Span> spans;
    std::vector<PqPage> pages;
    int64_t row0 = 0, raw_total = 0, dev_total = 0;
    bool any_comp = false, any_dict = false;
    for (const PqRowGroup &g : pf->groups) {
        const PqColumnChunk &cc = g.cols[i];
        if (cc.codec != 0 && cc.codec != 1) return fail(BOWGPU_ERR_UNSUPPORTED, "parquet: codec %d of column '%s' (UNCOMPRESSED and SNAPPY are read)", cc.codec, sc.name.c_str());
        // the chunk starts at its dictionary page when it has one (some writers leave dictionary_page_offset unset: the first
        // page header tells)
        const int64_t chunk_off = (cc.dictionary_page_offset >= 4 && cc.dictionary_page_offset < cc.data_page_offset) ? cc.dictionary_page_offset
                                                                                                                     : cc.data_page_offset;
        if (chunk_off < 4 || cc.total_compressed < 0 || chunk_off + cc.total_compressed > pf->size)
            return fail(BOWGPU_ERR_ARG, "parquet: column chunk of '%s' lies outside the file", sc.name.c_str());
        const uint8_t *chunk = pf->map + chunk_off;
        int64_t dict_off = -1;   // offset of this chunk's dictionary values (scratch buffer if it was compressed, else the chunk bytes)
        int32_t dict_count = 0, dict_in_raw = 0;
        const size_t chunk_len = (size_t)cc.total_compressed;
        const int64_t base = dev_total;
        size_t p = 0;
        int64_t vals = 0;
        while (p < chunk_len && vals < cc.num_values) {
            PageHdr h;
            if (!parse_page_header(chunk + p, chunk_len - p, &h)) return fail(BOWGPU_ERR_ARG, "parquet: malformed page header in column '%s'", sc.name.c_str());
            p += h.hdr_len;
            if (h.comp_size < 0 || p + (size_t)h.comp_size > chunk_len) return fail(BOWGPU_ERR_ARG, "parquet: page of column '%s' runs past its chunk", sc.name.c_str());
            // plausibility of the sizes a page header claims.  A DATA page holds at most 8 value bytes + level bits per value; a
            // DICTIONARY page carries its count in its own header field (DataPageHeader.num_values stays 0 there), is PLAIN 8-byte
            // values, and may well exceed 1 MiB (parquet-cpp only falls back to PLAIN once the dictionary has grown past its limit)
            const bool dict_hdr = h.type == 2;
            if (h.raw_size < 0 || h.num_values < 0 ||
                (!dict_hdr && (int64_t)h.raw_size > (int64_t)16 * h.num_values + (1 << 20)) ||
                (dict_hdr && (h.dict_values < 0 || (int64_t)h.dict_values > (int64_t)1 << 27 || (int64_t)h.raw_size != (int64_t)8 * h.dict_values)))
                return fail(BOWGPU_ERR_ARG, "parquet: implausible page header in column '%s'", sc.name.c_str());
            if (h.type == 2) {  // the chunk's dictionary: PLAIN values, decompressed like a page
                if ((h.dict_encoding != 0 && h.dict_encoding != 2) || dict_off >= 0 || (int64_t)h.raw_size != (int64_t)h.dict_values * 8)
                    return fail(BOWGPU_ERR_UNSUPPORTED, "parquet: dictionary page of column '%s' is not PLAIN 8-byte values", sc.name.c_str());
                PqPage pg;
                memset(&pg, 0, sizeof pg);
                pg.src_off = base + (int64_t)p;
                pg.comp_size = h.comp_size;
                pg.raw_size = h.raw_size;
                pg.compressed = cc.codec == 1 ? 1 : 0;
                pg.kind = 2;
                if (pg.compressed) {
                    pg.raw_off = raw_total;
                    raw_total += ((int64_t)h.raw_size + 15) & ~(int64_t)15;
                    any_comp = true;
                } else {
                    pg.raw_off = pg.src_off;  // a stored dictionary is read in place
                }
                dict_off = pg.raw_off;
                dict_in_raw = pg.compressed;
                dict_count = h.dict_values;
                pages.push_back(pg);
            } else if (h.type == 0 || (h.type == 3 && h.v2)) {
                const bool dict_page = h.encoding == 2 || h.encoding == 8;
                if (h.encoding != 0 && !dict_page) return fail(BOWGPU_ERR_UNSUPPORTED, "parquet: value encoding %d in column '%s' (PLAIN and dictionary are read)", h.encoding, sc.name.c_str());
                if (dict_page && dict_off < 0) return fail(BOWGPU_ERR_ARG, "parquet: column '%s' has dictionary-encoded pages but no dictionary", sc.name.c_str());
                if (optional && !h.v2 && h.def_encoding != 3) return fail(BOWGPU_ERR_UNSUPPORTED, "parquet: definition-level encoding %d in column '%s' (RLE is read)", h.def_encoding, sc.name.c_str());
                const int32_t lv_bytes = h.v2 ? h.def_len + h.rep_len : 0;
                if (h.v2 && (h.def_len < 0 || h.rep_len != 0 || lv_bytes > h.comp_size || lv_bytes > h.raw_size))
                    return fail(BOWGPU_ERR_ARG, "parquet: implausible level sizes in a v2 page of column '%s'", sc.name.c_str());
                PqPage pg;
                memset(&pg, 0, sizeof pg);
                pg.kind = dict_page ? 1 : 0;
                pg.v2 = h.v2 ? 1 : 0;
                pg.lv_off = base + (int64_t)p + h.rep_len;
                pg.lv_len = h.def_len;
                pg.dict_off = dict_off;
                pg.dict_in_raw = dict_in_raw;
                pg.dict_count = dict_count;
                pg.idx_off = row0 + vals;  // (one slot per row is always enough)
                any_dict = any_dict || dict_page;
                pg.src_off = base + (int64_t)p + lv_bytes;
                pg.comp_size = h.comp_size - lv_bytes;
                pg.raw_size = h.raw_size - lv_bytes;
                pg.num_values = h.num_values;
                pg.compressed = (cc.codec == 1 && (!h.v2 || h.v2_compressed)) ? 1 : 0;
                pg.row0 = row0 + vals;
                pg.raw_off = pg.compressed ? raw_total : pg.src_off;
                if (pg.compressed) { raw_total += ((int64_t)h.raw_size + 15) & ~(int64_t)15; any_comp = true; }
                pages.push_back(pg);
                vals += h.num_values;
            } else if (h.type == 3) {
                return fail(BOWGPU_ERR_UNSUPPORTED, "parquet: page type %d in column '%s' (data pages v1 / v2 are read)", h.type, sc.name.c_str());
            }  // (index pages are skipped)
            p += (size_t)h.comp_size;
        }
        if (vals != cc.num_values || vals != g.num_rows) return fail(BOWGPU_ERR_ARG, "parquet: column '%s' holds %lld values for %lld rows", sc.name.c_str(), (long long)vals, (long long)g.num_rows);
        spans.push_back({chunk_off, (int64_t)chunk_len, base});
        dev_total += (int64_t)chunk_len;
        row0 += g.num_rows;
    }
    if (row0 != n) return fail(BOWGPU_ERR_ARG, "parquet: row groups hold %lld rows, the footer says %lld", (long long)row0, (long long)n);

    // ---- upload + decode
    void *d_bytes, *d_pages, *d_raw = nullptr;
    BG_TRY(ctx_pool(c, kPoolInterp + 0, (size_t)dev_total + 32, &d_bytes));
    BG_TRY(ctx_pool(c, kPoolInterp + 1, pages.size() * sizeof(PqPage) + 32, &d_pages));
    if (any_comp) BG_TRY(ctx_pool(c, kPoolInterp + 2, (size_t)raw_total + 32, &d_raw));
    for (const Span &sp : spans)
        BG_TRY(copy_h2d(c, reinterpret_cast<char *>(d_bytes) + sp.dev_off, pf->map + sp.file_off, (size_t)sp.len));
    BG_HIP(hipMemcpyAsync(d_pages, pages.data(), pages.size() * sizeof(PqPage), hipMemcpyHostToDevice, c->stream));
    void *dscr;
    BG_TRY(ctx_scratch(c, 8192, &dscr));
    uint32_t *status = reinterpret_cast<uint32_t *>(dscr);
    unsigned long long *dcnt = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(dscr) + 1024);
    BG_HIP(hipMemsetAsync(status, 0, 64, c->stream));
    BG_HIP(hipMemsetAsync(dcnt, 0, 8, c->stream));
    BG_HIP(hipMemsetAsync(dout.validity, 0, (size_t)(((n + 7) >> 3) + 3) & ~(size_t)3, c->stream));
    void *d_idx = nullptr;
    if (any_dict) BG_TRY(ctx_pool(c, kPoolInterp + 3, (size_t)n * 4 + 32, &d_idx));
    BG_TRY(launch_parquet_decode(c, reinterpret_cast<const uint8_t *>(d_bytes), reinterpret_cast<const PqPage *>(d_pages), (int64_t)pages.size(),
                                 any_comp, reinterpret_cast<uint8_t *>(d_raw), optional, any_dict, reinterpret_cast<uint32_t *>(d_idx),
                                 reinterpret_cast<uint64_t *>(dout.values),
                                 reinterpret_cast<uint32_t *>(dout.validity), dcnt, status));
    uint32_t hstat = 0;
    uint64_t hcnt = 0;
    BG_HIP(hipMemcpyAsync(&hstat, status, 4, hipMemcpyDeviceToHost, c->stream));
    BG_HIP(hipMemcpyAsync(&hcnt, dcnt, 8, hipMemcpyDeviceToHost, c->stream));
    BG_HIP(hipStreamSynchronize(c->stream));
    if (hstat) return fail(BOWGPU_ERR_ARG, "parquet: malformed page data in column '%s' (status %u)", sc.name.c_str(), hstat);
    BG_TRY(devout_finish(c, &dout, n, otype, n - (int64_t)hcnt));
    BG_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
