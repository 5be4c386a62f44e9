// Mirror of the reference's t/testseqprofile.cpp (host only, no GPU).
#include <cstdlib>
#include <iostream>
#include "../hx_host.h"
using namespace historian;

int main(int argc, char** argv) {
  if (argc != 3) {
    std::cout << "Usage: " << argv[0] << " <alphabet> <sequence>\n";
    exit(EXIT_FAILURE);
  }
  FastSeq fs;
  fs.seq = argv[2];
  Profile prof(1, string(argv[1]), fs, 0);
  prof.writeJson(std::cout);
  exit(EXIT_SUCCESS);
}
