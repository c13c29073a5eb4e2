// cli_addnh.cpp — `mmannot-amd-addnh`: companion stream filter with the behaviour of the reference's addNH
// (addNH.cpp:31-93): every run of consecutive SAM lines that share their first column gets "\tNH:i:<run length>"
// appended; empty lines and '@' header lines pass through where they stand.  Host-only tool, not on the count path.
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

static void usage() {
  std::cerr << "Usage: addNH [options]\n"
            << "\t-i file: input SAM file (default: stdin)\n"
            << "\t-o file: output SAM file (default: stdout)\n"
            << "\t-h: this help" << std::endl;
}

int main(int argc, char **argv) {
  std::ifstream fin;
  std::ofstream fout;
  bool have_in = false, have_out = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a(argv[i]);
    if (a.empty()) continue;
    if (a == "-i" && i + 1 < argc) {
      fin.open(argv[++i]);
      if (!fin) { std::cerr << "Error: Cannot open input file '" << argv[i] << "'." << std::endl; return 1; }
      have_in = true;
    } else if (a == "-o" && i + 1 < argc) {
      fout.open(argv[++i]);
      if (!fout) { std::cerr << "Error: Cannot open output file '" << argv[i] << "'." << std::endl; return 1; }
      have_out = true;
    } else if (a == "-h") { usage(); return 0; }
    else { std::cerr << "Error: wrong parameter '" << a << "'.\nExiting." << std::endl; usage(); return 1; }
  }
  std::istream &in = have_in ? fin : std::cin;
  std::ostream &out = have_out ? fout : std::cout;
  std::vector<std::string> run;   // the alignment lines of the current read name
  std::string run_name, line;
  auto flush_run = [&]() {
    for (const std::string &l : run) out << l << "\tNH:i:" << run.size() << "\n";
    run.clear();
  };
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '@') { out << line << "\n"; continue; }   // printed at once, even inside a run
    const std::string name = line.substr(0, line.find('\t'));
    if (name != run_name) { flush_run(); run_name = name; }
    run.push_back(line);
  }
  flush_run();
  return 0;
}
