// water_cube_main.cpp — the reference's WATER_CUBE_TEST driver (src/main.cu:39-99,192-216) on the MI355X engine, positional
// arguments (see examples/raytracedicom_main.cpp for the reference's flag surface).
//
//   g++ -std=c++17 -O2 -Iinclude examples/water_cube_main.cpp -Lraytracedicom_amd -lrtd_hip -Wl,-rpath,$PWD/raytracedicom_amd -o water_cube
//   ./water_cube <LUT directory/> [output directory] [edge=256] [layers=20] [gpu_id=0]
#include "water_cube_plan.hpp"

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: %s <LUT dir/> [out dir] [edge] [layers] [gpu_id]\n", argv[0]); return 2; }
    const std::string lutDir = argv[1], outDir = argc > 2 ? argv[2] : ".";
    const unsigned int n = argc > 3 ? std::atoi(argv[3]) : 256;
    const unsigned int nLayers = argc > 4 ? std::atoi(argv[4]) : 20;
    const int gpuId = argc > 5 ? std::atoi(argv[5]) : 0;
    try {
        runWaterCube(lutDir, outDir, n, nLayers, std::vector<int>{gpuId});
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
