// Parses a mesh file with the C++ mirror's load_obj / load_stl (include/rpt.hpp, src/io.rs) and prints the
// triangle array: count, then 18 doubles per triangle.  Host-only (no GPU): used by tests/test_host_api.py to
// check the C++ loaders against the Python ones.   usage: mesh_dump <file> obj|stl
#include <fstream>
#include <iostream>
#include <string>

#include "rpt.hpp"

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) return 3;
    try {
        rpt::Shape m = std::string(argv[2]) == "obj" ? rpt::load_obj(f) : rpt::load_stl(f);
        std::cout.precision(17);
        std::cout << m.tris.size() / 18 << "\n";
        for (double v : m.tris) std::cout << v << "\n";
    } catch (const rpt::Error& e) {
        std::cerr << e.what() << "\n";
        return 1;
    }
    return 0;
}
