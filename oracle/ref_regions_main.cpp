// oracle/ref_regions_main.cpp -- TEST INFRASTRUCTURE: the reference's own region reader as a command.
//
//   ref_regions <bed> <max_regions> <chrom_limit or -> <order: 0|1>
//
// Calls readRegions / orderRegions of the reference (src/region.cpp, compiled from where it lies, see
// oracle/Makefile) and prints one line per region: chrom, start, stop, motif, name, period, period_str
// (tab-separated), then "lines <n>" is NOT available (readRegions only logs it), so the log goes to stdout after a
// line "--log--".  A malformed file ends the process inside the reference (printErrorAndDie: message on
// stderr, exit status 1) -- which is exactly what tests/test_io_formats.py compares against.
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "region.h"

int main(int argc, char** argv) {
  if (argc != 5) { std::cerr << "usage: ref_regions <bed> <max_regions> <chrom_limit|-> <order>\n"; return 2; }
  const std::string limit = (std::string(argv[3]) == "-") ? "" : argv[3];
  std::vector<Region> regions;
  std::ostringstream log;
  readRegions(argv[1], (uint32_t)std::strtoul(argv[2], nullptr, 10), limit, regions, log);
  if (std::atoi(argv[4])) orderRegions(regions);
  for (const Region& r : regions)
    std::cout << r.chrom() << "\t" << r.start() << "\t" << r.stop() << "\t" << r.motif() << "\t" << r.name() << "\t" << r.period() << "\t" << r.period_str() << "\n";
  std::cout << "--log--\n" << log.str();
  return 0;
}
