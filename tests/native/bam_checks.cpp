// CPU check of the host program's BAM writer (dart_amd/csrc/host/bam_writer.h), no GPU involved: reads a SAM file, writes it as BAM the
// way `dart -bo` does (header lines -> header, every other line -> add_sam_text), prints the record / refused counts.
// Test infrastructure: tests/test_host_text.py builds it with g++ -lz, runs it on the reference-generated golden SAM files and decodes the result.
#include "bam_writer.h"
#include <fstream>
#include <sstream>
int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: bam_checks in.sam out.bam threads\n"); return 2; }
    std::ifstream in(argv[1], std::ios::binary);
    std::stringstream ss; ss << in.rdbuf();
    const std::string text = ss.str();
    std::string header; std::vector<std::string> names; std::vector<int64_t> lens;
    size_t p = 0;
    while (p < text.size() && text[p] == '@') {
        const size_t e = text.find('\n', p);
        const std::string line = text.substr(p, e - p);
        header += line + "\n";
        if (line.compare(0, 3, "@SQ") == 0) {
            const size_t a = line.find("SN:") + 3, b = line.find('\t', a), c = line.find("LN:") + 3;
            names.push_back(line.substr(a, b - a)); lens.push_back(atoll(line.c_str() + c));
        }
        p = e + 1;
    }
    BamWriter w;
    if (!w.open(argv[2], header, names, lens, atoi(argv[3]))) return 1;
    // in pieces of awkward sizes that end on line boundaries, as the host program's formatter threads deliver them
    // alternately one piece (add_sam_text) and a group of pieces converted side by side (add_sam_chunks)
    size_t chunk = 7; int turn = 0;
    std::vector<std::pair<const char *, size_t>> group;
    while (p < text.size()) {
        size_t e = std::min(text.size(), p + chunk);
        e = text.find('\n', e - 1);
        e = e == std::string::npos ? text.size() : e + 1;
        if (turn % 5 == 0) w.add_sam_text(text.data() + p, e - p);
        else { group.emplace_back(text.data() + p, e - p); if (turn % 5 == 4) { w.add_sam_chunks(group); group.clear(); } }
        p = e; chunk = chunk * 3 + 11; if (chunk > (1u << 22)) chunk = 7;
        turn++;
    }
    if (!group.empty()) w.add_sam_chunks(group);
    if (!w.close()) return 1;
    printf("records=%lld refused=%lld\n", w.records(), w.refused());
    return 0;
}
