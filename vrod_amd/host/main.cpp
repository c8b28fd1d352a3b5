// main.cpp -- the `vrod` CLI with the reference's flags (src/main.rs:10-34):
//   -i/--init-database PATH   -n/--init-database-name NAME   -d/--database DIR
//   -c/--collection NAME      -e/--execute COMMAND           -a/--command-arg ARG
// (-g/--generate-embeddings is the reference's dev tool around fastembed; not part of the path.)
// Upstream parses -d/-c/-e/-a but never reads them (main.rs:64-76 is commented out); here they run
// through CommandBuilder::build exactly as that commented block sketches.
#include <cstdio>
#include <cstring>
#include <optional>
#include <string>

#include "command.hpp"

using namespace vrod_host;

static void usage() {
    std::puts("Usage: vrod [OPTIONS]\n\nOptions:\n"
              "  -i, --init-database <PATH>\n  -n, --init-database-name <NAME>\n  -d, --database <DIR>\n"
              "  -c, --collection <COLLECTION_NAME>\n  -e, --execute <COMMAND>\n  -a, --command-arg <COMMAND_ARG>\n"
              "  -h, --help");
}

int main(int argc, char** argv) {
    std::optional<std::string> init_database, init_database_name, database, collection, execute, command_arg;
    if (argc < 2) {  // #[command(arg_required_else_help(true))]
        usage();
        return 2;
    }
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&](std::optional<std::string>& dst) -> bool {
            if (i + 1 >= argc) { std::fprintf(stderr, "error: a value is required for '%s'\n", a.c_str()); return false; }
            dst = argv[++i];
            return true;
        };
        bool ok = true;
        if (a == "-i" || a == "--init-database") ok = val(init_database);
        else if (a == "-n" || a == "--init-database-name") ok = val(init_database_name);
        else if (a == "-d" || a == "--database") ok = val(database);
        else if (a == "-c" || a == "--collection") ok = val(collection);
        else if (a == "-e" || a == "--execute") ok = val(execute);
        else if (a == "-a" || a == "--command-arg") ok = val(command_arg);
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else { std::fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str()); return 2; }
        if (!ok) return 2;
    }
    try {
        if (init_database) {
            if (!init_database_name) {  // ArgsError::MissingInitDatabaseNameFlag (main.rs:36-40)
                std::fprintf(stderr, "Error: Missing '--init_database_name' flag with argument for '--init_database' flag.\n");
                return 1;
            }
            Database::new_(*init_database, *init_database_name);
            return 0;
        }
        if (!execute) return 0;  // upstream falls through to Ok(())
        DbHandle db = Database::load(database ? *database : std::string("."));
        CommandBuilder builder(db);
        std::unique_ptr<Command> cmd = builder.build(collection, *execute, command_arg);
        cmd->execute();
        return 0;
    } catch (const CommandBuilderError& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    } catch (const IoError& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    } catch (const DeviceError& e) {
        std::fprintf(stderr, "Error: device (status %d): %s\n", e.status, e.what());
        return 1;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());
        return 1;
    }
}
