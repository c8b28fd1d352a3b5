// database.hpp -- host mirror of vRod's `Database` (reference src/database/mod.rs:6-22,
// src/database/setup.rs:3-26), extended with what the reference leaves as TODO:
// "//TODO collections" (mod.rs:8) and `Database::load` (mod.rs:19-21, todo!()).
//
// On-disk layout (the reference fixes the first three entries, setup.rs:17-23):
//   <path>/<name>/                 database directory, AlreadyExists error if present
//   <path>/<name>/vr_config        database config (reference: empty file)
//   <path>/<name>/vr_wal           write-ahead log placeholder (reference: empty file)
//   <path>/<name>/<collection>/vr_config     key=value: dim, metric, dtype, count
//   <path>/<name>/<collection>/vr_vectors    raw little-endian fp32 rows, insertion order
//   <path>/<name>/<collection>/vr_payloads   one payload line per row (the `;word` part
//                                            of the reference's text format, embeddings.rs:55-61)
// Vectors live in HBM behind a vrod_index handle (include/vrod.h), loaded on first use.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vrod.h"

namespace vrod_host {

// mirrors std::io::Error as used by Database::new (kind + message)
struct IoError : std::runtime_error {
    enum Kind { AlreadyExists, NotFound, InvalidData, Other } kind;
    IoError(Kind k, const std::string& m) : std::runtime_error(m), kind(k) {}
};

// errors coming back through the C ABI (status + vrod_last_error())
struct DeviceError : std::runtime_error {
    int status;
    DeviceError(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

struct CollectionConfig {
    uint32_t dim = 0;                       // 0 = fixed by the first inserted vector
    int metric = VROD_METRIC_COSINE;
    int dtype = VROD_DTYPE_F32;
    uint64_t count = 0;
};

class Collection {
public:
    Collection(std::string dir, std::string name);
    ~Collection();
    Collection(const Collection&) = delete;
    Collection& operator=(const Collection&) = delete;

    const std::string& name() const { return name_; }
    const CollectionConfig& config() const { return cfg_; }
    void set_config(const CollectionConfig& c) { cfg_ = c; }

    void save_config() const;
    void trim_to_count();   // drop rows / payload lines beyond `count` (leftovers of an uncommitted insert)
    void load_config();
    // append rows (n x dim fp32) + payloads to disk and, if resident, to the device index
    void insert(const std::vector<float>& rows, uint32_t dim, const std::vector<std::string>& payloads);
    // nq x dim queries -> nq x k (ids, scores); loads the collection into HBM on first use
    void search(const std::vector<float>& queries, uint32_t nq, uint32_t k, std::vector<uint64_t>& ids,
                std::vector<float>& scores);
    std::string payload(uint64_t id);

private:
    void ensure_resident();
    std::string dir_, name_;
    CollectionConfig cfg_;
    vrod_index* index_ = nullptr;
    std::vector<std::string> payload_cache_;
    bool payloads_loaded_ = false;
    bool trimmed_ = false;
};

class Database {
public:
    // reference: Database::new(path, name) -> io::Result<Self>  (mod.rs:13-17)
    static std::shared_ptr<Database> new_(const std::string& path, const std::string& name);
    // reference: Database::load(path)  (mod.rs:19-21, todo!() upstream)
    static std::shared_ptr<Database> load(const std::string& path);

    const std::string& path() const { return path_; }
    Collection& create_collection(const std::string& name, const CollectionConfig& cfg);
    void drop_collection(const std::string& name);
    std::vector<std::string> list_collections() const;
    Collection& collection(const std::string& name);

private:
    explicit Database(std::string path) : path_(std::move(path)) {}
    std::string path_;  // the database directory itself (<path>/<name> of new_)
    std::map<std::string, std::unique_ptr<Collection>> collections_;
};

// reference: setup::create_database_directory (setup.rs:3-26)
void create_database_directory(const std::string& path, const std::string& name);

}  // namespace vrod_host
