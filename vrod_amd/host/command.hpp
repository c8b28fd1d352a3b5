// command.hpp -- host mirror of vRod's command layer, name for name:
//   trait Command { fn execute(&self); }                 reference src/command/types.rs:5-7
//   the 11 command structs + UnrecognizedCommand           types.rs:9-154 (all bodies empty upstream)
//   CommandBuilder::new / CommandBuilder::build            src/command/builder.rs:17-81
//   CommandBuilderError::UnrecognizedCommand               builder.rs:10-15
// The structs keep the reference's field names (db, collection_name, arg, target, command).
// `db` is a shared_ptr: the reference's Rc<RefCell<Database>> (types.rs:10) -- shared,
// single-threaded ownership.  execute() takes nothing and returns nothing, as upstream; a
// command that produces output keeps it on `self` (results / message) and prints it.
//
// Only the scan path has a device body: SEARCHSIMILAR (the hot path), plus the minimum that
// feeds it (CREATE, DROP, LISTCOLLECTIONS, INSERT, BULKINSERT).  TRUNCATEWAL, UPDATE, DELETE,
// SEARCH and REINDEX stay the empty stubs they are upstream (out of scope).
#pragma once
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "database.hpp"

namespace vrod_host {

typedef std::shared_ptr<Database> DbHandle;  // Rc<RefCell<Database>>

class Command {
public:
    virtual ~Command() = default;
    virtual void execute() = 0;  // fn execute(&self)
};

struct CreateCollectionCommand : Command {
    DbHandle db;
    std::optional<std::string> collection_name;  // builder passes `arg` here (builder.rs:30-33)
    void execute() override;
};
struct DropCollectionCommand : Command {
    DbHandle db;
    std::optional<std::string> collection_name;
    void execute() override;
};
struct ListCollectionsCommand : Command {
    DbHandle db;
    std::vector<std::string> names;  // filled by execute
    void execute() override;
};
struct TruncateWalCommand : Command {
    DbHandle db;
    std::optional<std::string> target;
    void execute() override {}  // stub upstream (types.rs:49-54); out of scope
};
struct InsertCommand : Command {
    DbHandle db;
    std::optional<std::string> collection_name;
    std::optional<std::string> arg;  // one line of the reference's text format: f,f,...,f;payload
    void execute() override;
};
struct BulkInsertCommand : Command {
    DbHandle db;
    std::optional<std::string> collection_name;
    std::optional<std::string> arg;  // path of a file: text lines `f,f,...;payload`, or raw fp32 `*.f32:DIM`
    uint64_t inserted = 0;
    void execute() override;
};
struct UpdateCommand : Command {
    DbHandle db;
    std::optional<std::string> collection_name, arg;
    void execute() override {}  // stub upstream
};
struct DeleteCommand : Command {
    DbHandle db;
    std::optional<std::string> collection_name, arg;
    void execute() override {}  // stub upstream
};
struct SearchCommand : Command {
    DbHandle db;
    std::optional<std::string> collection_name, arg;
    void execute() override {}  // stub upstream
};
// THE HOT PATH (types.rs:121-132).  arg syntax (build-defined):
//     [k=K;] v,v,...,v [; v,v,...,v ...]        queries inline, `;`-separated
//     [k=K;] @FILE                              one query per line, `f,f,...[;payload]`
struct SearchSimilarCommand : Command {
    DbHandle db;
    std::optional<std::string> collection_name;
    std::optional<std::string> arg;
    uint32_t k = 10, nq = 0;
    std::vector<uint64_t> ids;     // nq x k, best first (VROD_ID_NONE past the collection size)
    std::vector<float> scores;     // nq x k
    bool quiet = false;            // do not print (tests)
    void execute() override;
};
struct ReindexCommand : Command {
    DbHandle db;
    std::optional<std::string> collection_name;
    void execute() override {}  // stub upstream; a brute-force index has nothing to rebuild
};
struct UnrecognizedCommand : Command {
    std::string command;
    void execute() override {}
};

// #[error("Unrecognized command: {0}")]
struct CommandBuilderError : std::runtime_error {
    std::string command;
    explicit CommandBuilderError(const std::string& c) : std::runtime_error("Unrecognized command: " + c), command(c) {}
};

class CommandBuilder {
public:
    explicit CommandBuilder(DbHandle db) : db_(std::move(db)) {}  // CommandBuilder::new
    // Result<Box<dyn Command>, CommandBuilderError>: returns the command or throws the error
    std::unique_ptr<Command> build(std::optional<std::string> collection, const std::string& command,
                                   std::optional<std::string> arg);

private:
    DbHandle db_;
};

// text format of the reference's embeddings file (src/utils/embeddings.rs:55-61)
bool parse_vector_line(const std::string& line, std::vector<float>& values, std::string& payload);

}  // namespace vrod_host
