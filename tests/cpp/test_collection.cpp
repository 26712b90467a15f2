// The reference's search tests (collection_test.go) against the C++ host mirror
// (include/syzgy_collection.hpp).  Built and run by tests/test_gpu_cpp_host.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#include "syzgy_collection.hpp"

using namespace syzgydb;

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "CHECK failed at %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::exit(1);                                                            \
        }                                                                            \
    } while (0)

static void TestExhaustiveSearch()  // collection_test.go:549-612
{
    CollectionOptions o;
    o.DistanceMethod = Euclidean;
    o.DimensionCount = 3;
    auto c = Collection::NewCollection(o);
    c->AddDocument(1, {1.0, 2.0, 3.0}, "doc1");
    c->AddDocument(2, {4.0, 5.0, 6.0}, "doc2");
    c->AddDocument(3, {7.0, 8.0, 9.0}, "doc3");
    SearchArgs a;
    a.Vector = {1.0, 2.0, 3.0};
    a.Precision = "exact";
    a.K = 3;
    const SearchResults r = c->Search(a);
    CHECK(r.Results.size() == 3);
    std::set<uint64_t> ids;
    for (const auto &x : r.Results) ids.insert(x.ID);
    CHECK(ids == (std::set<uint64_t>{1, 2, 3}));
    CHECK(r.PercentSearched == 100.0);
    CHECK(r.Results[0].Distance == 0.0 && r.Results[1].Distance == 5.196152422706632);  // :12-21 KAT
    CHECK(r.Results[2].Distance == 10.392304845413264 && r.Results[0].Metadata == "doc1");
}

static void TestCollectionSearch()  // collection_test.go:283-382
{
    CollectionOptions o;
    o.DistanceMethod = Euclidean;
    o.DimensionCount = 2;
    {
        auto empty = Collection::NewCollection(o);
        SearchArgs a;
        a.Vector = {50, 50};
        a.K = 5;
        CHECK(empty->Search(a).Results.empty());
    }
    auto c = Collection::NewCollection(o);
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> u(0, 100);
    for (int i = 0; i < 10; i++) c->AddDocument(i, {u(rng), u(rng)}, "metadata");
    SearchArgs a;
    a.Vector = {50, 50};
    a.K = 5;
    CHECK(!c->Search(a).Results.empty());
    a.K = 3;
    CHECK(c->Search(a).Results.size() <= 3);
    SearchArgs r;
    r.Vector = {50, 50};
    r.Radius = 10;
    for (const auto &x : c->Search(r).Results) CHECK(x.Distance <= 10);
    SearchArgs f;
    f.Vector = {50, 50};
    f.K = 5;
    f.Filter = [](uint64_t id, const std::string &) { return id % 2 == 0; };
    const SearchResults fr = c->Search(f);
    CHECK(fr.Results.size() == 5);
    for (const auto &x : fr.Results) CHECK(x.ID % 2 == 0);
    CHECK(fr.PercentSearched == 100.0);
}

static void TestCrudAndQuantization()
{
    CollectionOptions o;
    o.DistanceMethod = Euclidean;
    o.DimensionCount = 3;
    o.Quantization = 4;  // collection_test.go:614-667
    auto c = Collection::NewCollection(o);
    for (int i = 0; i < 10; i++) c->AddDocument(i, {0.1 * i, 0.05 * i, 0.9 - 0.1 * i}, "metadata");
    SearchArgs a;
    a.Vector = {0.3, 0.2, 0.5};
    a.K = 5;
    CHECK(c->Search(a).Results.size() == 5);
    const Document d = c->GetDocument(3);
    CHECK(std::fabs(d.Vector[0] - 0.3) < 0.07);  // 4-bit grid
    c->removeDocument(3);
    bool threw = false;
    try {
        c->GetDocument(3);
    } catch (const std::runtime_error &) {
        threw = true;
    }
    CHECK(threw && c->GetDocumentCount() == 9);
    c->AddDocument(5, {0.3, 0.2, 0.5}, "moved");  // rewriting an id replaces its vector
    a.K = 1;
    CHECK(c->Search(a).Results[0].ID == 5 && c->Search(a).Results[0].Metadata == "moved");
    threw = false;
    try {
        c->AddDocument(99, {1.0}, "");  // dimension mismatch panics in the reference
    } catch (const std::invalid_argument &) {
        threw = true;
    }
    CHECK(threw);
    SearchArgs l;  // listing mode: sorted string ids
    l.Limit = 3;
    const auto lr = c->Search(l).Results;
    CHECK(lr.size() == 3 && lr[0].ID == 0 && lr[1].ID == 1 && lr[2].ID == 2);
}

static void TestOpenCollectionFile(const char *path, uint64_t expect_best, double expect_dist)
{
    CollectionOptions o;
    o.Name = path;
    auto c = Collection::NewCollection(o);  // options come from the header record
    SearchArgs a;
    a.Vector.assign(c->GetOptions().DimensionCount, 0.25);
    a.K = 3;
    a.Precision = "exact";
    const SearchResults r = c->Search(a);
    CHECK(r.Results.size() == 3 && r.PercentSearched == 100.0);
    std::printf("file: dim=%d q=%d metric=%d docs=%d best=%llu dist=%.17g\n", c->GetOptions().DimensionCount,
                c->GetOptions().Quantization, c->GetOptions().DistanceMethod, c->GetDocumentCount(),
                (unsigned long long)r.Results[0].ID, r.Results[0].Distance);
    CHECK(r.Results[0].ID == expect_best && r.Results[0].Distance == expect_dist);
}

static void TestComputeAverageDistance()  // collection_test.go:105-142
{
    CollectionOptions o;
    o.DistanceMethod = Euclidean;
    o.DimensionCount = 3;
    auto c = Collection::NewCollection(o);
    uint64_t s = 12345;
    auto rnd = [&s]() {  // any generator: the reference's draws come from its own math/rand
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(s >> 11) / 9007199254740992.0;
    };
    for (int i = 0; i < 100; i++) c->AddDocument((uint64_t)i, {rnd() * 100, rnd() * 100, rnd() * 100}, "metadata");
    const double avg = c->computeAverageDistance(50, [&](int n) { return (int)(rnd() * n); });
    CHECK(avg > 0);
    // two fixed documents: the mean over one pair is their reference distance (:12-21 KAT)
    auto d = Collection::NewCollection(o);
    d->AddDocument(1, {1, 2, 3}, "");
    d->AddDocument(2, {4, 5, 6}, "");
    int turn = 0;
    CHECK(d->computeAverageDistance(1, [&](int) { return turn++ & 1; }) == 5.196152422706632);
    CHECK(d->computeAverageDistance(0, [&](int) { return 0; }) == 0.0);
    CHECK(d->computeAverageDistance(3, [&](int) { return 0; }) == 0.0);  // id1 == id2 every time
}

static void TestSearchBatch()  // the batch surface: same answers as Search by Search
{
    CollectionOptions o;
    o.DistanceMethod = Cosine;
    o.DimensionCount = 32;
    o.Quantization = 8;
    auto c = Collection::NewCollection(o);
    uint64_t s = 99;
    auto rnd = [&s]() {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(s >> 11) / 9007199254740992.0 * 2 - 1;
    };
    for (int i = 0; i < 3000; i++) {
        std::vector<double> v(32);
        for (double &x : v) x = rnd();
        c->AddDocument((uint64_t)(i + 10), v, i % 3 ? "odd" : "three");
    }
    std::vector<SearchArgs> batch(20);
    for (size_t i = 0; i < batch.size(); i++) {
        batch[i].Vector.resize(32);
        for (double &x : batch[i].Vector) x = rnd();
        batch[i].K = 5;
        batch[i].Precision = "exact";
        if (i % 2) batch[i].Filter = [](uint64_t, const std::string &m) { return m == "three"; };
    }
    const std::vector<SearchResults> got = c->SearchBatch(batch);
    CHECK(got.size() == batch.size());
    for (size_t i = 0; i < batch.size(); i++) {
        const SearchResults one = c->Search(batch[i]);
        CHECK(one.Results.size() == got[i].Results.size() && got[i].PercentSearched == 100.0);
        for (size_t j = 0; j < one.Results.size(); j++) {
            CHECK(one.Results[j].ID == got[i].Results[j].ID && one.Results[j].Distance == got[i].Results[j].Distance);
            CHECK(!(i % 2) || got[i].Results[j].Metadata == "three");
        }
    }
    szg_stats st;
    CHECK(szg_get_stats(c->handle(), &st) == SZG_OK && st.mq_queries >= batch.size());  // the batch shared a sweep
}

int main(int argc, char **argv)
{
    TestSearchBatch();
    TestExhaustiveSearch();
    TestComputeAverageDistance();
    TestCollectionSearch();
    TestCrudAndQuantization();
    if (argc >= 4) TestOpenCollectionFile(argv[1], std::strtoull(argv[2], nullptr, 10), std::strtod(argv[3], nullptr));
    std::printf("CPP_HOST_OK\n");
    return 0;
}
