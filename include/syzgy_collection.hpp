// syzgy_collection.hpp -- C++ host-side mirror of the reference's Collection API
// for the search path, header-only, on top of the C ABI (syzgy_scan.h,
// syzgy_pager.h).  Same names, argument meaning and error behaviour as
// collection.go: CollectionOptions (:31-48), Document (:100-110), SearchArgs
// (:140-158), SearchResult(s) (:115-135), NewCollection (:224-314), AddDocument
// (:427-457), GetDocument (:463-484), UpdateDocument (:490-509), removeDocument
// (:511-521), Search (:569-711).  Where the reference panics (log.Panicf) this
// throws; where it returns an error it throws std::runtime_error.
//
// What differs is only WHERE the exact scan runs: the hot loop (:672-684) is one
// call into libsyzgy_scan.so.  Storage stays out of scope: documents added here
// live in host memory; an existing collection FILE is opened read-only through
// the pager (the header record's options override the caller's, :241-252).
// Any Precision is answered by the exact scan (the LSH path is unchanged Go).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "syzgy_pager.h"
#include "syzgy_scan.h"

namespace syzgydb {

enum { Euclidean = 0, Cosine = 1 };  // collection.go:186-189

struct CollectionOptions {
    std::string Name;
    int DistanceMethod = Euclidean;
    int DimensionCount = 0;
    int Quantization = 0;  // 0 -> 64 (collection.go:254-256)
};

struct Document {
    uint64_t ID = 0;
    std::vector<double> Vector;
    std::string Metadata;
};

struct SearchResult {
    uint64_t ID = 0;
    std::string Metadata;
    double Distance = 0;
};

struct SearchResults {
    std::vector<SearchResult> Results;
    double PercentSearched = 0;
};

using FilterFn = std::function<bool(uint64_t id, const std::string &metadata)>;

struct SearchArgs {
    std::vector<double> Vector;
    FilterFn Filter;
    int K = 0;
    double Radius = 0;
    int Offset = 0;
    int Limit = 0;
    std::string Precision;
};

class Collection {
public:
    // NewCollection: opens options.Name when it is an existing non-empty collection
    // file, otherwise starts an empty in-memory collection with the given options.
    static std::unique_ptr<Collection> NewCollection(CollectionOptions options,
                                                     const std::vector<int> &devices = {})
    {
        std::unique_ptr<Collection> c(new Collection());
        szg_pager *pg = nullptr;
        bool from_file = false;
        if (!options.Name.empty()) {
            std::ifstream f(options.Name, std::ios::binary | std::ios::ate);
            from_file = f.good() && f.tellg() > 0;
        }
        if (from_file) {
            if (szg_pager_open(&pg, options.Name.c_str(), 0) != SZG_OK)
                throw std::runtime_error("failed to read header");  // collection.go:243-245
            szg_pager_options(pg, &options.DimensionCount, &options.Quantization, &options.DistanceMethod);
        } else if (options.Quantization == 0) {
            options.Quantization = 64;
        }
        if (options.DistanceMethod != Euclidean && options.DistanceMethod != Cosine) {
            if (pg) szg_pager_close(pg);
            throw std::runtime_error("unsupported distance method");  // collection.go:282
        }
        if (szg_row_bytes(options.Quantization, options.DimensionCount) < 0) {
            if (pg) szg_pager_close(pg);
            throw std::invalid_argument("Unsupported quantization level");  // panic, collection.go:809
        }
        c->opts_ = options;
        const int rc = szg_index_create(&c->ix_, options.DimensionCount, options.Quantization,
                                        options.DistanceMethod, devices.empty() ? nullptr : devices.data(),
                                        (int)devices.size());
        if (rc != SZG_OK) {
            if (pg) szg_pager_close(pg);
            throw std::runtime_error(std::string("szg_index_create: ") + szg_last_error());
        }
        if (pg) {
            const uint64_t n = szg_pager_count(pg);
            int rc2 = szg_pager_load(pg, c->ix_);
            std::vector<uint64_t> ids(n);
            if (rc2 == SZG_OK) rc2 = szg_pager_ids(pg, ids.data());
            for (uint64_t r = 0; r < n && rc2 == SZG_OK; r++) {
                const uint8_t *m = nullptr;
                uint64_t len = 0;
                rc2 = szg_pager_metadata(pg, r, &m, &len);
                c->row_of_[ids[r]] = r;
                c->id_of_.push_back(ids[r]);
                c->live_.push_back(true);
                c->meta_.emplace_back(reinterpret_cast<const char *>(m), (size_t)len);
            }
            szg_pager_close(pg);
            if (rc2 != SZG_OK) throw std::runtime_error("failed to iterate records");
        }
        return c;
    }

    ~Collection() { Close(); }
    Collection(const Collection &) = delete;
    Collection &operator=(const Collection &) = delete;

    const CollectionOptions &GetOptions() const { return opts_; }
    int GetDocumentCount() const { return (int)row_of_.size(); }

    std::vector<uint64_t> GetAllIDs() const
    {
        std::vector<uint64_t> ids;
        for (const auto &kv : row_of_) ids.push_back(kv.first);
        return ids;  // std::map: ascending, like the sort in collection.go:339
    }

    void AddDocument(uint64_t id, const std::vector<double> &vector, const std::string &metadata)
    {
        if ((int)vector.size() != opts_.DimensionCount)  // log.Panicf, collection.go:432-434
            throw std::invalid_argument("vector size does not match the expected number of dimensions");
        auto it = row_of_.find(id);
        if (it != row_of_.end()) {
            // WriteRecord of an existing id replaces the record
            check(szg_index_overwrite_f64(ix_, it->second, vector.data()), "szg_index_overwrite_f64");
            meta_[it->second] = metadata;
            return;
        }
        check(szg_index_append_f64(ix_, vector.data(), 1), "szg_index_append_f64");
        row_of_[id] = id_of_.size();
        id_of_.push_back(id);
        live_.push_back(true);
        meta_.push_back(metadata);
    }

    Document GetDocument(uint64_t id)
    {
        auto it = row_of_.find(id);
        if (it == row_of_.end()) throw std::runtime_error("record not found");  // spanfile.go:516
        const int dim = opts_.DimensionCount, q = opts_.Quantization;
        std::vector<uint8_t> b((size_t)szg_row_bytes(q, dim));
        check(szg_index_read_rows(ix_, it->second, 1, b.data()), "szg_index_read_rows");
        Document d;
        d.ID = id;
        d.Metadata = meta_[it->second];
        d.Vector.resize(dim);
        for (int i = 0; i < dim; i++) d.Vector[i] = decode(b.data(), i, q);  // collection.go:768-794
        return d;
    }

    void UpdateDocument(uint64_t id, const std::string &newMetadata)
    {
        auto it = row_of_.find(id);
        if (it == row_of_.end()) throw std::runtime_error("record not found");
        meta_[it->second] = newMetadata;
    }

    void removeDocument(uint64_t id)
    {
        auto it = row_of_.find(id);
        if (it == row_of_.end()) throw std::runtime_error("record not found");
        check(szg_index_tombstone(ix_, it->second), "szg_index_tombstone");
        live_[it->second] = false;
        meta_[it->second].clear();
        row_of_.erase(it);
    }

    // collection.go:348-400; intn(n) plays rand.Intn
    double computeAverageDistance(int samples, const std::function<int(int)> &intn)
    {
        if (samples <= 0) return 0.0;
        const std::vector<uint64_t> ids = GetAllIDs();
        if (ids.size() < 2) return 0.0;
        std::vector<uint64_t> a, b;
        for (int i = 0; i < samples; i++) {
            const uint64_t id1 = ids[(size_t)intn((int)ids.size())];
            const uint64_t id2 = ids[(size_t)intn((int)ids.size())];
            if (id1 == id2) continue;
            a.push_back(row_of_[id1]);
            b.push_back(row_of_[id2]);
        }
        if (a.empty()) return 0.0;
        std::vector<double> d(a.size());
        check(szg_pair_distances(ix_, a.data(), b.data(), a.size(), d.data()), "szg_pair_distances");
        double totalDistance = 0.0;
        for (double x : d) totalDistance += x;
        return totalDistance / (double)a.size();
    }

    // collection.go:569-711
    SearchResults Search(const SearchArgs &args)
    {
        SearchResults ret;
        const size_t numRecords = row_of_.size();
        size_t pointsSearched = 0;
        if (args.Radius == 0 && args.K == 0) {
            // listing mode (:633-669): sorted *string* id order, Offset / Limit
            std::vector<std::pair<std::string, uint64_t>> ids;
            for (const auto &kv : row_of_) ids.emplace_back(std::to_string(kv.first), kv.first);
            std::sort(ids.begin(), ids.end());
            for (const auto &p : ids) {
                const uint64_t row = row_of_[p.second];
                if (args.Filter && !args.Filter(p.second, meta_[row])) continue;
                pointsSearched++;
                if (args.Offset > 0 && (int)pointsSearched <= args.Offset) continue;
                ret.Results.push_back(SearchResult{p.second, meta_[row], 0.0});
                if (args.Limit > 0 && (int)ret.Results.size() >= args.Limit) break;
            }
        } else {
            if ((int)args.Vector.size() != opts_.DimensionCount)  // undefined in the reference (:814, :823)
                throw std::invalid_argument("query length does not match the collection's dimension");
            std::vector<uint64_t> allow;
            const uint64_t total = szg_index_rows(ix_);
            if (args.Filter && numRecords) {  // :592-594, one bit per row
                allow.assign((total + 63) / 64, 0);
                for (const auto &kv : row_of_)
                    if (args.Filter(kv.first, meta_[kv.second])) allow[kv.second / 64] |= 1ull << (kv.second % 64);
            }
            const uint64_t *ap = allow.empty() ? nullptr : allow.data();
            std::vector<uint64_t> rows;
            std::vector<double> dist;
            size_t n = 0;
            if (numRecords == 0) {
                n = 0;
            } else if (args.Radius > 0) {  // K is ignored (:598-605)
                uint64_t cap = 1u << 16, totalHits = 0;  // a truncated call costs a second sweep
                for (;;) {
                    rows.assign(cap, 0);
                    dist.assign(cap, 0);
                    const int rc = szg_search_radius(ix_, args.Vector.data(), args.Radius, ap, rows.data(),
                                                     dist.data(), cap, &totalHits);
                    if (rc == SZG_E_TRUNCATED) {
                        cap = totalHits;
                        continue;
                    }
                    check(rc, "szg_search_radius");
                    break;
                }
                n = (size_t)totalHits;
            } else {
                rows.assign(args.K, 0);
                dist.assign(args.K, 0);
                int32_t count = 0;
                check(szg_search_topk(ix_, args.Vector.data(), 1, args.K, ap, rows.data(), dist.data(), &count),
                      "szg_search_topk");
                n = (size_t)count;
            }
            for (size_t i = 0; i < n; i++)
                ret.Results.push_back(SearchResult{id_of_[rows[i]], meta_[rows[i]], dist[i]});
            pointsSearched = numRecords;  // counted before the filter (:589)
        }
        ret.PercentSearched = numRecords ? (double)pointsSearched / (double)numRecords * 100 : 0;
        return ret;
    }

    // Exact top-k Searches with one K for a caller that holds many queries (not in the reference, whose REST endpoint
    // is single-query, rest.go:371-487): ONE szg_search_topk call -- batches of two or more share sweeps of the
    // corpus on the matrix cores -- each query with its own Filter.  Same results as Search called one by one;
    // anything else (radius, listing, mixed K) is answered Search by Search.
    std::vector<SearchResults> SearchBatch(const std::vector<SearchArgs> &args)
    {
        std::vector<SearchResults> out;
        const size_t numRecords = row_of_.size();
        bool same = !args.empty() && numRecords > 0;
        for (const SearchArgs &a : args)
            same = same && a.Radius == 0 && a.K > 0 && a.K == args[0].K && (int)a.Vector.size() == opts_.DimensionCount;
        if (!same) {
            for (const SearchArgs &a : args) out.push_back(Search(a));
            return out;
        }
        const int nq = (int)args.size(), k = args[0].K, dim = opts_.DimensionCount;
        const uint64_t total = szg_index_rows(ix_);
        const size_t words = (size_t)((total + 63) / 64);
        std::vector<double> q((size_t)nq * dim);
        std::vector<uint64_t> allow;
        bool any = false;
        for (const SearchArgs &a : args) any = any || (bool)a.Filter;
        if (any) allow.assign((size_t)nq * words, ~0ull);
        for (int i = 0; i < nq; i++) {
            std::copy(args[i].Vector.begin(), args[i].Vector.end(), q.begin() + (size_t)i * dim);
            if (!args[i].Filter) continue;
            std::fill(allow.begin() + (size_t)i * words, allow.begin() + (size_t)(i + 1) * words, 0ull);
            for (const auto &kv : row_of_)
                if (args[i].Filter(kv.first, meta_[kv.second])) allow[(size_t)i * words + kv.second / 64] |= 1ull << (kv.second % 64);
        }
        std::vector<uint64_t> rows((size_t)nq * k);
        std::vector<double> dist((size_t)nq * k);
        std::vector<int32_t> count(nq);
        check(szg_search_topk(ix_, q.data(), nq, k, any ? allow.data() : nullptr, rows.data(), dist.data(), count.data()),
              "szg_search_topk");
        for (int i = 0; i < nq; i++) {
            SearchResults r;
            for (int j = 0; j < count[i]; j++) {
                const uint64_t row = rows[(size_t)i * k + j];
                r.Results.push_back(SearchResult{id_of_[row], meta_[row], dist[(size_t)i * k + j]});
            }
            r.PercentSearched = 100;
            out.push_back(r);
        }
        return out;
    }

    void Close()
    {
        if (ix_) {
            szg_index_destroy(ix_);
            ix_ = nullptr;
        }
    }

    szg_index *handle() { return ix_; }

private:
    Collection() = default;

    static void check(int rc, const char *where)
    {
        if (rc != SZG_OK) throw std::runtime_error(std::string(where) + ": " + szg_last_error());
    }

    // decodeVector + dequantize, collection.go:768-794 / quantization.go:25-36
    static double decode(const uint8_t *data, int i, int q)
    {
        uint64_t v = 0;
        switch (q) {
        case 4: v = (i % 2 == 0) ? (uint64_t)(data[i / 2] >> 4) : (uint64_t)(data[i / 2] & 0x0F); break;
        case 8: v = data[i]; break;
        case 16: v = ((uint64_t)data[i * 2] << 8) | data[i * 2 + 1]; break;
        case 32: for (int b = 0; b < 4; b++) v = (v << 8) | data[i * 4 + b]; break;
        default: for (int b = 0; b < 8; b++) v = (v << 8) | data[i * 8 + b]; break;
        }
        if (q == 32) {
            const uint32_t u = (uint32_t)v;
            float f;
            std::memcpy(&f, &u, 4);
            return (double)f;
        }
        if (q == 64) {
            double d;
            std::memcpy(&d, &v, 8);
            return d;
        }
        const double maxInt = (double)((1ull << q) - 1);
        return ((double)v / maxInt) * 2 - 1;
    }

    CollectionOptions opts_;
    szg_index *ix_ = nullptr;
    std::map<uint64_t, uint64_t> row_of_;  // id -> row
    std::vector<uint64_t> id_of_;          // row -> id
    std::vector<bool> live_;
    std::vector<std::string> meta_;        // row -> metadata
};

}  // namespace syzgydb
