// oracle/ref_csv_harness.cpp -- TEST INFRASTRUCTURE.  Drives the REFERENCE'S OWN third-party CSV parser
// (/root/reference/include/csv.hpp, compiled where it lies; nothing is copied into this repository) with exactly the
// reader set-up of utils::loadPointCloudCSV (/root/reference/src/utils.cpp:19-37 "ouster", :63-73 generic) and prints
// the parsed rows, so that the row-skipping / field semantics restated in icet_amd/csrc/icet_io.cpp and oracle/scan_io.py
// are pinned against the real parser.  The loader function itself cannot be compiled here (it returns Eigen::MatrixXf and
// Eigen is absent), which is why only its parser calls are exercised.  Built by `make -C oracle ref` into oracle/_ref/.
//   usage: csv_ref <file> ouster|xyz      -> one "a b c" line per row (integers in mm / floats with %.9g)
#include "csv.hpp"

#include <cstdio>
#include <fstream>
#include <string>

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    std::ifstream file(argv[1]);
    if (!file.is_open()) return 3;
    const std::string mode = argv[2];
    if (mode == "ouster") {
        csv::CSVReader reader(file, csv::CSVFormat().header_row(1).trim({}));
        csv::CSVRow row; reader.read_row(row);
        csv::CSVRow second; reader.read_row(second);
        for (csv::CSVRow& r : reader) std::printf("%d %d %d\n", r[8].get<int>(), r[9].get<int>(), r[10].get<int>());
    } else {
        csv::CSVReader reader(file, csv::CSVFormat().delimiter('\t'));
        for (csv::CSVRow& r : reader) std::printf("%.9g %.9g %.9g\n", std::stof(r[0].get<>()), std::stof(r[1].get<>()), std::stof(r[2].get<>()));
    }
    return 0;
}
