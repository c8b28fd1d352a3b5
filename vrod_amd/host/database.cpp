// database.cpp -- see database.hpp.  Host logic only: every scan goes through the C ABI of
// libvrod_hip.so (include/vrod.h); there is no CPU search path here.
#include "database.hpp"

#include <sys/stat.h>
#include <sys/types.h>
#include <dirent.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace vrod_host {

namespace {

bool path_exists(const std::string& p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}
bool is_dir(const std::string& p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
std::string join(const std::string& a, const std::string& b) {
    if (a.empty()) return b;
    return a.back() == '/' ? a + b : a + "/" + b;
}
void make_dir(const std::string& p) {
    if (::mkdir(p.c_str(), 0777) != 0) throw IoError(IoError::Other, "cannot create directory '" + p + "': " + std::strerror(errno));
}
void touch(const std::string& p) {
    std::ofstream f(p, std::ios::binary | std::ios::trunc);
    if (!f) throw IoError(IoError::Other, "cannot create file '" + p + "'");
}
void check(int rc, const char* what) {
    if (rc != VROD_OK) throw DeviceError(rc, std::string(what) + ": " + vrod_last_error());
}
const char* metric_name(int m) { return m == VROD_METRIC_L2 ? "l2" : "cosine"; }
const char* dtype_name(int d) { return d == VROD_DTYPE_BF16 ? "bf16" : "f32"; }

}  // namespace

// ---------------------------------------------------------------- setup.rs:3-26
void create_database_directory(const std::string& path, const std::string& name) {
    const std::string database_dir = join(path, name);
    if (path_exists(database_dir))
        throw IoError(IoError::AlreadyExists,
                      "Directory with the name '" + name + "' already exists in '" + path + "'");
    make_dir(database_dir);
    touch(join(database_dir, "vr_config"));
    touch(join(database_dir, "vr_wal"));
}

// ---------------------------------------------------------------- Collection
Collection::Collection(std::string dir, std::string name) : dir_(std::move(dir)), name_(std::move(name)) {}

Collection::~Collection() {
    if (index_) vrod_index_destroy(index_);
}

// vr_config is the commit point of an insert: written to a temporary file and renamed over the old
// one, so a reader sees either the old count or the new one, never a torn file.
void Collection::save_config() const {
    const std::string path = join(dir_, "vr_config"), tmp = path + ".tmp";
    {
        std::ofstream f(tmp, std::ios::trunc);
        if (!f) throw IoError(IoError::Other, "cannot write vr_config of collection '" + name_ + "'");
        f << "dim=" << cfg_.dim << "\nmetric=" << metric_name(cfg_.metric) << "\ndtype=" << dtype_name(cfg_.dtype)
          << "\ncount=" << cfg_.count << "\n";
        f.flush();
        if (!f) throw IoError(IoError::Other, "short write to vr_config of collection '" + name_ + "'");
    }
    if (::rename(tmp.c_str(), path.c_str()) != 0)
        throw IoError(IoError::Other, "cannot replace vr_config of collection '" + name_ + "'");
}

// Rows beyond `count` are leftovers of an insert that never reached its commit point (a failed or
// short write, a crash before save_config): cut vr_vectors back to count rows and vr_payloads to
// count lines, so that the next append lands where vr_config says the collection ends.  Once per
// process: a completed insert leaves the files consistent.
void Collection::trim_to_count() {
    if (trimmed_) return;
    const std::string vpath = join(dir_, "vr_vectors"), ppath = join(dir_, "vr_payloads");
    struct stat sb;
    const uint64_t want = cfg_.count * (uint64_t)cfg_.dim * 4ull;
    if (::stat(vpath.c_str(), &sb) == 0 && (uint64_t)sb.st_size > want && ::truncate(vpath.c_str(), (off_t)want) != 0)
        throw IoError(IoError::Other, "cannot trim vr_vectors of '" + name_ + "'");
    if (::stat(ppath.c_str(), &sb) == 0 && sb.st_size > 0) {
        std::ifstream f(ppath, std::ios::binary);
        uint64_t lines = 0, off = 0;
        std::vector<char> buf(1 << 20);
        bool cut = false;
        while (f && !cut) {
            f.read(buf.data(), (std::streamsize)buf.size());
            const std::streamsize got = f.gcount();
            for (std::streamsize i = 0; i < got; ++i) {
                if (lines == cfg_.count) { cut = true; off += (uint64_t)i; break; }
                if (buf[(size_t)i] == '\n') ++lines;
            }
            if (!cut) off += (uint64_t)got;
        }
        if (cut && off < (uint64_t)sb.st_size && ::truncate(ppath.c_str(), (off_t)off) != 0)
            throw IoError(IoError::Other, "cannot trim vr_payloads of '" + name_ + "'");
    }
    trimmed_ = true;
}

void Collection::load_config() {
    std::ifstream f(join(dir_, "vr_config"));
    if (!f) throw IoError(IoError::NotFound, "collection '" + name_ + "' has no vr_config");
    std::string line;
    while (std::getline(f, line)) {
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        const std::string k = line.substr(0, eq), v = line.substr(eq + 1);
        if (k == "dim") cfg_.dim = (uint32_t)std::stoul(v);
        else if (k == "metric") cfg_.metric = v == "l2" ? VROD_METRIC_L2 : VROD_METRIC_COSINE;
        else if (k == "dtype") cfg_.dtype = v == "bf16" ? VROD_DTYPE_BF16 : VROD_DTYPE_F32;
        else if (k == "count") cfg_.count = std::stoull(v);
    }
}

// VROD_DEVICES=0,1,2,...: the GPUs a collection is sharded over (one multi-device handle,
// include/vrod.h); unset = the first device.
static int create_index(vrod_index** out, uint32_t dim, int dtype, int metric) {
    std::vector<int> devs;
    if (const char* e = std::getenv("VROD_DEVICES")) {
        std::string tok;
        for (const char* p = e;; ++p) {
            if (*p == ',' || *p == '\0') {
                if (!tok.empty()) devs.push_back(std::atoi(tok.c_str()));
                tok.clear();
                if (*p == '\0') break;
            } else tok.push_back(*p);
        }
    }
    return vrod_index_create(out, dim, dtype, metric, devs.empty() ? nullptr : devs.data(), (int)devs.size());
}

void Collection::ensure_resident() {
    if (index_) return;
    if (cfg_.dim == 0) throw IoError(IoError::InvalidData, "collection '" + name_ + "' is empty");
    check(create_index(&index_, cfg_.dim, cfg_.dtype, cfg_.metric), "vrod_index_create");
    if (cfg_.count == 0) return;
    check(vrod_index_reserve(index_, cfg_.count), "vrod_index_reserve");
    std::ifstream f(join(dir_, "vr_vectors"), std::ios::binary);
    if (!f) throw IoError(IoError::NotFound, "collection '" + name_ + "' has no vr_vectors");
    const uint64_t chunk = std::max<uint64_t>(1, (64ull << 20) / (cfg_.dim * 4ull));  // stream 64 MB at a time
    std::vector<float> buf(chunk * cfg_.dim);
    for (uint64_t done = 0; done < cfg_.count; done += chunk) {
        const uint64_t m = std::min(chunk, cfg_.count - done);
        f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)(m * cfg_.dim * 4));
        if ((uint64_t)f.gcount() != m * cfg_.dim * 4)
            throw IoError(IoError::InvalidData, "vr_vectors of '" + name_ + "' is shorter than vr_config says");
        check(vrod_index_add(index_, buf.data(), m), "vrod_index_add");
    }
}

void Collection::insert(const std::vector<float>& rows, uint32_t dim, const std::vector<std::string>& payloads) {
    if (dim == 0 || rows.size() % dim != 0) throw IoError(IoError::InvalidData, "ragged vectors");
    const uint64_t n = rows.size() / dim;
    if (n == 0) return;
    if (cfg_.dim == 0) cfg_.dim = dim;  // the first vector fixes the collection's dimension
    if (dim != cfg_.dim)
        throw IoError(IoError::InvalidData, "vector dimension " + std::to_string(dim) + " != collection dimension " +
                                                std::to_string(cfg_.dim));
    for (const std::string& p : payloads)
        if (p.find('\n') != std::string::npos || p.find('\r') != std::string::npos)
            throw IoError(IoError::InvalidData, "a payload must be one line (vr_payloads holds one line per vector)");
    // device first (it validates NaN/Inf), then disk
    if (index_ || cfg_.count > 0) ensure_resident();
    if (!index_) check(create_index(&index_, cfg_.dim, cfg_.dtype, cfg_.metric), "vrod_index_create");
    trim_to_count();
    const uint64_t count0 = cfg_.count;
    check(vrod_index_add(index_, rows.data(), n), "vrod_index_add");
    // disk: vectors, payloads, then vr_config (the commit point).  A failure before the commit leaves
    // orphans beyond `count`, which the next process trims; this process must not go on with a device
    // copy the disk does not have.
    try {
        {
            std::ofstream f(join(dir_, "vr_vectors"), std::ios::binary | std::ios::app);
            if (!f) throw IoError(IoError::Other, "cannot append to vr_vectors of '" + name_ + "'");
            f.write(reinterpret_cast<const char*>(rows.data()), (std::streamsize)(rows.size() * 4));
            f.flush();
            if (!f) throw IoError(IoError::Other, "short write to vr_vectors of '" + name_ + "'");
        }
        {
            std::ofstream f(join(dir_, "vr_payloads"), std::ios::app);
            if (!f) throw IoError(IoError::Other, "cannot append to vr_payloads of '" + name_ + "'");
            for (uint64_t i = 0; i < n; ++i) f << (i < payloads.size() ? payloads[i] : std::string()) << "\n";
            f.flush();
            if (!f) throw IoError(IoError::Other, "short write to vr_payloads of '" + name_ + "'");
        }
        cfg_.count += n;
        save_config();
    } catch (...) {
        // the device holds rows the disk does not: drop the handle (reloaded from disk on next use)
        cfg_.count = count0;   // save_config throws only while the old vr_config is still in place
        vrod_index_destroy(index_);
        index_ = nullptr;
        trimmed_ = false;
        payloads_loaded_ = false;
        payload_cache_.clear();
        throw;
    }
    if (payloads_loaded_)
        for (uint64_t i = 0; i < n; ++i) payload_cache_.push_back(i < payloads.size() ? payloads[i] : std::string());
}

void Collection::search(const std::vector<float>& queries, uint32_t nq, uint32_t k, std::vector<uint64_t>& ids,
                        std::vector<float>& scores) {
    ensure_resident();
    if (queries.size() != (size_t)nq * cfg_.dim)
        throw IoError(IoError::InvalidData, "query dimension != collection dimension " + std::to_string(cfg_.dim));
    ids.assign((size_t)nq * k, VROD_ID_NONE);
    scores.assign((size_t)nq * k, 0.f);
    check(vrod_search(index_, queries.data(), nq, k, ids.data(), scores.data()), "vrod_search");
}

std::string Collection::payload(uint64_t id) {
    if (!payloads_loaded_) {
        // only the first `count` lines are payloads of committed rows: lines behind them are orphans of an insert that
        // never reached its commit point (trim_to_count removes them from the file at the next insert)
        std::ifstream f(join(dir_, "vr_payloads"));
        std::string line;
        payload_cache_.clear();
        while (payload_cache_.size() < cfg_.count && std::getline(f, line)) payload_cache_.push_back(line);
        payloads_loaded_ = true;
    }
    return id < payload_cache_.size() ? payload_cache_[id] : std::string();
}

// ---------------------------------------------------------------- Database
std::shared_ptr<Database> Database::new_(const std::string& path, const std::string& name) {
    create_database_directory(path, name);
    return std::shared_ptr<Database>(new Database(join(path, name)));
}

std::shared_ptr<Database> Database::load(const std::string& path) {
    if (!is_dir(path) || !path_exists(join(path, "vr_config")))
        throw IoError(IoError::NotFound, "'" + path + "' is not a vRod database directory (no vr_config)");
    std::shared_ptr<Database> db(new Database(path));
    DIR* d = ::opendir(path.c_str());
    if (!d) throw IoError(IoError::Other, "cannot read '" + path + "'");
    while (dirent* e = ::readdir(d)) {
        const std::string n = e->d_name;
        if (n == "." || n == "..") continue;
        const std::string sub = join(path, n);
        if (is_dir(sub) && path_exists(join(sub, "vr_config"))) {
            auto c = std::make_unique<Collection>(sub, n);
            c->load_config();
            db->collections_[n] = std::move(c);
        }
    }
    ::closedir(d);
    return db;
}

Collection& Database::create_collection(const std::string& name, const CollectionConfig& cfg) {
    if (name.empty() || name.find('/') != std::string::npos || name == "vr_config" || name == "vr_wal")
        throw IoError(IoError::InvalidData, "bad collection name '" + name + "'");
    const std::string dir = join(path_, name);
    if (path_exists(dir)) throw IoError(IoError::AlreadyExists, "Collection '" + name + "' already exists");
    make_dir(dir);
    auto c = std::make_unique<Collection>(dir, name);
    c->set_config(cfg);
    c->save_config();
    touch(join(dir, "vr_vectors"));
    touch(join(dir, "vr_payloads"));
    Collection& ref = *c;
    collections_[name] = std::move(c);
    return ref;
}

void Database::drop_collection(const std::string& name) {
    auto it = collections_.find(name);
    if (it == collections_.end()) throw IoError(IoError::NotFound, "Collection '" + name + "' does not exist");
    collections_.erase(it);
    const std::string dir = join(path_, name);
    for (const char* f : {"vr_config", "vr_vectors", "vr_payloads"}) ::unlink(join(dir, f).c_str());
    ::rmdir(dir.c_str());
}

std::vector<std::string> Database::list_collections() const {
    std::vector<std::string> out;
    for (auto& kv : collections_) out.push_back(kv.first);
    return out;
}

Collection& Database::collection(const std::string& name) {
    auto it = collections_.find(name);
    if (it == collections_.end()) throw IoError(IoError::NotFound, "Collection '" + name + "' does not exist");
    return *it->second;
}

}  // namespace vrod_host
