// TEST INFRASTRUCTURE: drives helen_amd/csrc/h5emit.h directly -- one group of N scalar datasets whose names have
// every length modulo 8, nested two levels deep, plus an empty group -- so that the multi-level B-tree, the heap
// padding and the empty-group case can be read back through libhdf5 without writing gigabytes of predictions.
//   emit_many <file> <N>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "h5emit.h"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const int n = atoi(argv[2]);
    h5emit::File f;
    if (!f.open(argv[1])) return 3;
    std::vector<h5emit::Child> many;
    for (int i = 0; i < n; ++i) {
        // name i: decimal i followed by i % 9 letters -> lengths that hit every padding case
        std::string name = std::to_string(i) + std::string(i % 9, (char)('a' + i % 26));
        many.push_back({name, f.scalar_i64((int64_t)i * 3 - 7)});
    }
    std::vector<h5emit::Child> none;
    std::vector<h5emit::Child> mid{{"many", f.group(many)}, {"empty", f.group(none)}, {"answer", f.scalar_i64(42)}};
    std::vector<h5emit::Child> top{{"mid", f.group(mid)}};
    uint64_t bt = 0, hp = 0;
    const uint64_t root = f.group(top, &bt, &hp);
    return f.finish(root, bt, hp) ? 0 : 4;
}
