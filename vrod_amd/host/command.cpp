// command.cpp -- bodies of the commands on or feeding the scan path (see command.hpp).
#include "command.hpp"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace vrod_host {

namespace {
std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}
const std::string& need(const std::optional<std::string>& v, const char* what) {
    if (!v || v->empty()) throw IoError(IoError::InvalidData, std::string("missing ") + what);
    return *v;
}
bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
}
}  // namespace

// `f,f,...,f;payload` -- the payload is everything after the first ';' (may be absent)
bool parse_vector_line(const std::string& line, std::vector<float>& values, std::string& payload) {
    values.clear();
    payload.clear();
    const size_t semi = line.find(';');
    const std::string nums = trim(semi == std::string::npos ? line : line.substr(0, semi));
    if (semi != std::string::npos) payload = line.substr(semi + 1);
    if (nums.empty()) return false;
    const char* p = nums.c_str();
    while (*p) {
        char* end = nullptr;
        const float v = std::strtof(p, &end);  // accepts what Rust's f32::to_string writes, incl. NaN/inf
        if (end == p) return false;
        values.push_back(v);
        p = end;
        while (*p && std::isspace((unsigned char)*p)) ++p;
        if (*p == ',') ++p;
        else if (*p) return false;
    }
    return !values.empty();
}

// CREATE: arg = "NAME [metric=cosine|l2] [dtype=f32|bf16] [dim=N]"
void CreateCollectionCommand::execute() {
    std::istringstream in(need(collection_name, "collection name"));
    std::string name, tok;
    in >> name;
    CollectionConfig cfg;
    while (in >> tok) {
        const size_t eq = tok.find('=');
        const std::string k = tok.substr(0, eq), v = eq == std::string::npos ? "" : tok.substr(eq + 1);
        if (k == "metric") cfg.metric = (v == "l2" || v == "L2") ? VROD_METRIC_L2 : VROD_METRIC_COSINE;
        else if (k == "dtype") cfg.dtype = v == "bf16" ? VROD_DTYPE_BF16 : VROD_DTYPE_F32;
        else if (k == "dim") cfg.dim = (uint32_t)std::stoul(v);
        else throw IoError(IoError::InvalidData, "unknown CREATE option '" + tok + "'");
    }
    db->create_collection(name, cfg);
}

void DropCollectionCommand::execute() { db->drop_collection(need(collection_name, "collection name")); }

void ListCollectionsCommand::execute() {
    names = db->list_collections();
    for (auto& n : names) std::printf("%s\n", n.c_str());
}

void InsertCommand::execute() {
    Collection& c = db->collection(need(collection_name, "collection (-c)"));
    std::vector<float> v;
    std::string payload;
    if (!parse_vector_line(need(arg, "vector (-a)"), v, payload)) throw IoError(IoError::InvalidData, "cannot parse vector");
    c.insert(v, (uint32_t)v.size(), {payload});
}

void BulkInsertCommand::execute() {
    Collection& c = db->collection(need(collection_name, "collection (-c)"));
    std::string path = need(arg, "file (-a)");
    inserted = 0;
    const uint64_t batch_rows = 65536;
    // raw little-endian fp32: "FILE.f32:DIM"
    const size_t colon = path.rfind(':');
    if (colon != std::string::npos && ends_with(path.substr(0, colon), ".f32")) {
        const uint32_t dim = (uint32_t)std::stoul(path.substr(colon + 1));
        std::ifstream f(path.substr(0, colon), std::ios::binary);
        if (!f || dim == 0) throw IoError(IoError::NotFound, "cannot open '" + path + "'");
        std::vector<float> buf(batch_rows * dim);
        for (;;) {
            f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(buf.size() * 4));
            const uint64_t got = (uint64_t)f.gcount() / (4ull * dim);
            if (!got) break;
            std::vector<float> rows(buf.begin(), buf.begin() + got * dim);
            c.insert(rows, dim, {});
            inserted += got;
        }
        return;
    }
    std::ifstream f(path);
    if (!f) throw IoError(IoError::NotFound, "cannot open '" + path + "'");
    std::vector<float> rows, v;
    std::vector<std::string> payloads;
    std::string line, payload;
    uint32_t dim = 0;
    uint64_t lineno = 0;
    auto flush = [&]() {
        if (rows.empty()) return;
        c.insert(rows, dim, payloads);
        inserted += payloads.size();
        rows.clear();
        payloads.clear();
    };
    while (std::getline(f, line)) {
        ++lineno;
        if (trim(line).empty()) continue;
        if (!parse_vector_line(line, v, payload))
            throw IoError(IoError::InvalidData, path + ":" + std::to_string(lineno) + ": cannot parse vector");
        if (dim == 0) dim = (uint32_t)v.size();
        if (v.size() != dim)
            throw IoError(IoError::InvalidData, path + ":" + std::to_string(lineno) + ": ragged vector (" +
                                                    std::to_string(v.size()) + " values, expected " + std::to_string(dim) + ")");
        rows.insert(rows.end(), v.begin(), v.end());
        payloads.push_back(payload);
        if (payloads.size() == batch_rows) flush();
    }
    flush();
}

void SearchSimilarCommand::execute() {
    Collection& c = db->collection(need(collection_name, "collection (-c)"));
    std::string a = trim(need(arg, "query (-a)"));
    // optional leading "k=K;"
    if (a.rfind("k=", 0) == 0) {
        const size_t semi = a.find(';');
        k = (uint32_t)std::stoul(a.substr(2, semi == std::string::npos ? std::string::npos : semi - 2));
        a = semi == std::string::npos ? "" : trim(a.substr(semi + 1));
    }
    std::vector<float> queries, v;
    std::string payload;
    nq = 0;
    uint32_t dim = 0;
    auto add = [&](const std::string& text, const std::string& where) {
        if (trim(text).empty()) return;
        if (!parse_vector_line(text, v, payload)) throw IoError(IoError::InvalidData, where + ": cannot parse query");
        if (dim == 0) dim = (uint32_t)v.size();
        if (v.size() != dim) throw IoError(IoError::InvalidData, where + ": ragged query");
        queries.insert(queries.end(), v.begin(), v.end());
        ++nq;
    };
    if (!a.empty() && a[0] == '@') {
        std::ifstream f(a.substr(1));
        if (!f) throw IoError(IoError::NotFound, "cannot open '" + a.substr(1) + "'");
        std::string line;
        uint64_t lineno = 0;
        while (std::getline(f, line)) add(line, a.substr(1) + ":" + std::to_string(++lineno));
    } else {
        // inline: queries separated by ';' (a query itself has no payload here)
        size_t pos = 0;
        while (pos <= a.size()) {
            const size_t semi = a.find(';', pos);
            add(a.substr(pos, semi == std::string::npos ? std::string::npos : semi - pos), "-a");
            if (semi == std::string::npos) break;
            pos = semi + 1;
        }
    }
    if (nq == 0) throw IoError(IoError::InvalidData, "no query vectors in -a");
    c.search(queries, nq, k, ids, scores);
    if (quiet) return;
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t i = 0; i < k; ++i) {
            const uint64_t id = ids[(size_t)q * k + i];
            if (id == VROD_ID_NONE) break;
            const std::string p = c.payload(id);
            std::printf("%u\t%u\t%llu\t%.9g\t%s\n", q, i, (unsigned long long)id, (double)scores[(size_t)q * k + i], p.c_str());
        }
}

// builder.rs:22-81 -- same names, same (collection, command, arg) routing, same error
std::unique_ptr<Command> CommandBuilder::build(std::optional<std::string> collection, const std::string& command,
                                               std::optional<std::string> arg) {
    std::string up = command;
    std::transform(up.begin(), up.end(), up.begin(), [](unsigned char ch) { return (char)std::toupper(ch); });
    DbHandle db = db_;
    if (up == "CREATE") { auto c = std::make_unique<CreateCollectionCommand>(); c->db = db; c->collection_name = arg; return c; }
    if (up == "DROP") { auto c = std::make_unique<DropCollectionCommand>(); c->db = db; c->collection_name = arg; return c; }
    if (up == "LISTCOLLECTIONS") { auto c = std::make_unique<ListCollectionsCommand>(); c->db = db; return c; }
    if (up == "TRUNCATEWAL") { auto c = std::make_unique<TruncateWalCommand>(); c->db = db; c->target = collection; return c; }
    if (up == "INSERT") { auto c = std::make_unique<InsertCommand>(); c->db = db; c->collection_name = collection; c->arg = arg; return c; }
    if (up == "BULKINSERT") { auto c = std::make_unique<BulkInsertCommand>(); c->db = db; c->collection_name = collection; c->arg = arg; return c; }
    if (up == "UPDATE") { auto c = std::make_unique<UpdateCommand>(); c->db = db; c->collection_name = collection; c->arg = arg; return c; }
    if (up == "DELETE") { auto c = std::make_unique<DeleteCommand>(); c->db = db; c->collection_name = collection; c->arg = arg; return c; }
    if (up == "SEARCH") { auto c = std::make_unique<SearchCommand>(); c->db = db; c->collection_name = collection; c->arg = arg; return c; }
    if (up == "SEARCHSIMILAR") { auto c = std::make_unique<SearchSimilarCommand>(); c->db = db; c->collection_name = collection; c->arg = arg; return c; }
    if (up == "REINDEX") { auto c = std::make_unique<ReindexCommand>(); c->db = db; c->collection_name = collection; return c; }
    throw CommandBuilderError(command);
}

}  // namespace vrod_host
