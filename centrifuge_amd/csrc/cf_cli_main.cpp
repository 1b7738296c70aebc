// cf_cli_main.cpp — `centrifuge-class`: main() around the library entry point, as the reference's
// centrifuge_main.cpp:34-68 (including its `-A <file>` mode: one argument string per line, run in turn).
#include <cstdio>
#include <cstring>
#include <fstream>
#include <malloc.h>
#include <sstream>
#include <string>
#include <vector>

extern "C" int centrifuge(int argc, const char **argv);

int main(int argc, const char **argv) {
    // the ingest pool allocates and frees a ~16 MiB structure-of-arrays chunk per parsed block: keep those
    // blocks on the heap (no mmap / munmap and page faults per chunk).  Process-wide, hence here and not in
    // the library entry point.
    mallopt(M_MMAP_THRESHOLD, 32 << 20);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    if (argc > 2 && std::strcmp(argv[1], "-A") == 0) {
        std::ifstream in(argv[2]);
        std::string line;
        int last = -1;
        while (std::getline(in, line)) {
            std::vector<std::string> args{argv[0]};
            std::istringstream ss(line);
            for (std::string t; ss >> t;) args.push_back(t);
            if (args.size() == 1) continue;
            std::vector<const char *> av;
            for (const auto &a : args) av.push_back(a.c_str());
            last = centrifuge((int)av.size(), av.data());
        }
        if (last == -1) { std::fprintf(stderr, "Warning: No arg strings parsed from %s\n", argv[2]); return 0; }
        return last;
    }
    return centrifuge(argc, argv);
}
