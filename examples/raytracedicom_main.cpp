// raytracedicom_main.cpp — the reference executable's command-line surface (src/config.cpp:13-51, src/main.cu:20-24,
// 199-216) on the MI355X engine.
//
//   g++ -std=c++17 -O2 -Iinclude -Iexamples examples/raytracedicom_main.cpp -Lraytracedicom_amd -lrtd_hip
//       -Wl,-rpath,$PWD/raytracedicom_amd -o raytracedicom
//   ./raytracedicom --water_cube --lut_dir /path/to/LUTs/ --output_directory out [--gpu_id 0] [--config_file run.ini]
//
// Flags of the reference, same names and meaning:
//   --gpu_id N              device to compute on (the reference parses it but never uses it, config.cpp:13-15; honoured here)
//   --ct_dir DIR            patient CT directory            } required unless --water_cube (as in a non-WATER_CUBE_TEST
//   --rtplan FILE           RT (ion) plan DICOM file        } build, config.cpp:18-39)
//   --beams NAME [NAME...]  beam names to include           }
//   --output_directory DIR  existing directory for dose.dat (required, config.cpp:41-46)
//   --config_file FILE      key = value file; command-line arguments override it (config.cpp:48-51)
// Added (the reference bakes these in at compile time: CMakeLists.txt:32-33 PHYS_DATA_DIRECTORY, :36 WATER_CUBE_TEST):
//   --lut_dir DIR           directory of the LUT text files (default: $RTD_LUT_DIR, then ./LUTs/)
//   --water_cube            run the reference's WATER_CUBE_TEST plan (main.cu:39-99) instead of reading DICOM input
//   --water_cube_edge N     voxels per cube edge (default 256)      --layers N   energy layers (default 20)
//   --spot_list FILE        with --water_cube: the plan is read from a spot-list text file (E x y fwhm_x fwhm_y meterset rows,
//                           gantry_angle / isocenter / source_dist keys) and turned into BeamSettings by include/rtd_plan.hpp
//                           (the step the reference's main.cu:150-197 stops before)
//   --dump_spot_list FILE   write the built-in water-cube plan as such a spot list and exit
// The parsed configuration is echoed as key=value lines like the reference's app.config_to_str (config.cpp:62).
// Without --water_cube the CT series and the RT Ion Plan are read by include/rtd_dicom.hpp (no ITK / GDCM), the named beams —
// one or several, where the reference throws "Multi-beam calculation not yet supported" (main.cu:117-120) — become
// BeamSettings through include/rtd_plan.hpp and their dose is written to dose.dat on the CT grid. The reference's own DICOM
// path stops after printing the plan, with zero spot weights (main.cu:105-190).
//   --start_depth MM / --tracer_steps N   override the tracer range that is otherwise fitted to the CT along each beam axis
#include <sys/stat.h>

#include <map>
#include <sstream>

#include "water_cube_plan.hpp"

namespace {

struct Config {
    unsigned short gpu_id = 0;
    std::vector<int> gpu_ids;       // extension: several devices -> beams are computed on several GPUs (rtd_plan_*)
    bool fine_grained_timing = false;   // the reference's FINE_GRAINED_TIMING build option, at run time
    std::string ct_dir, rtplan, output_directory, lut_dir, spot_list, dump_spot_list;
    std::vector<std::string> beams;
    bool water_cube = false;
    unsigned int water_cube_edge = 256, layers = 20;
    float start_depth = 0.0f;       // DICOM mode: gantry z of tracer step 0 (0 = just upstream of the CT)
    unsigned int tracer_steps = 0;  // DICOM mode: 0 = as many as it takes to cross the CT
};

bool isDir(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
bool isFile(const std::string& p) { struct stat st; return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
// values of one key: "a b", "a,b", ["a","b"]
std::vector<std::string> splitValues(std::string v) {
    v = trim(v);
    if (!v.empty() && v.front() == '[' && v.back() == ']') v = v.substr(1, v.size() - 2);
    std::vector<std::string> out;
    std::string cur; char quote = 0;
    for (char c : v) {
        if (quote) { if (c == quote) quote = 0; else cur += c; continue; }
        if (c == '"' || c == '\'') { quote = c; continue; }
        if (c == ',' || c == ' ' || c == '\t') { if (!cur.empty()) { out.push_back(cur); cur.clear(); } continue; }
        cur += c;
    }
    if (!cur.empty()) out.push_back(cur);
    return out;
}

[[noreturn]] void usageError(const std::string& msg) {
    std::cerr << msg << "\nRun with --help for more information.\n";
    std::exit(2);
}

void printHelp() {
    std::cout << "RayTraceDicom: Sub-second pencil beam dose calculation on GPU for adaptive proton therapy.\n"
                 "Usage: raytracedicom [OPTIONS]\n\nOptions:\n"
                 "  -h,--help                   Print this help message and exit\n"
                 "  --gpu_id UINT               ID of the GPU to use for simulation, starts from 0.\n"
                 "  --gpu_ids UINT ...          Several GPUs: the beams of the plan are computed on all of them (same dose, bit for bit).\n"
                 "  --fine_grained_timing       Per-field timing lines of the reference's FINE_GRAINED_TIMING build.\n"
                 "  --ct_dir TEXT               Patient CT directory. It must contain all the DICOM CT slices.\n"
                 "  --rtplan TEXT:FILE          Path of the RTPLAN DICOM file to read.\n"
                 "  --beams TEXT ...            All beam names to include in the calculation\n"
                 "  --output_directory TEXT:DIR REQUIRED Directory where output will be stored.\n"
                 "  --config_file TEXT          Specify a config file containing simulation parameters. The contents of the file\n"
                 "                              are overriden by command line arguments.\n"
                 "  --lut_dir TEXT:DIR          Directory of the LUT text files (default $RTD_LUT_DIR, then ./LUTs/).\n"
                 "  --water_cube                Run the reference's WATER_CUBE_TEST plan instead of reading DICOM input.\n"
                 "  --water_cube_edge UINT      Voxels per cube edge (256).\n"
                 "  --layers UINT               Energy layers of the water-cube plan (20).\n"
                 "  --start_depth FLOAT         DICOM input: gantry z (mm) of tracer step 0 (default: just upstream of the CT).\n"
                 "  --tracer_steps UINT         DICOM input: number of 1 mm tracer steps (default: enough to cross the CT).\n"
                 "  --spot_list TEXT:FILE       With --water_cube: plan from a spot-list text file (E x y fwhm_x fwhm_y meterset).\n"
                 "  --dump_spot_list TEXT       Write the built-in water-cube plan as a spot list and exit.\n";
}

unsigned long parseUInt(const std::string& key, const std::string& v, unsigned long maxV) {
    char* end = nullptr;
    const unsigned long x = std::strtoul(v.c_str(), &end, 10);
    if (v.empty() || *end != '\0' || v[0] == '-' || x > maxV) usageError("--" + key + ": could not convert '" + v + "' to an unsigned integer");
    return x;
}

// one (key, values) assignment, from the file or from the command line
void assign(Config& c, const std::string& key, const std::vector<std::string>& vals, std::map<std::string, bool>& seen) {
    auto one = [&]() -> const std::string& {
        if (vals.size() != 1) usageError("--" + key + ": 1 required TEXT missing");
        return vals[0];
    };
    if (key == "gpu_id") c.gpu_id = (unsigned short)parseUInt(key, one(), 65535);
    else if (key == "gpu_ids") { if (vals.empty()) usageError("--gpu_ids: At least 1 required"); c.gpu_ids.clear(); for (const auto& v : vals) c.gpu_ids.push_back((int)parseUInt(key, v, 65535)); }
    else if (key == "fine_grained_timing") {
        if (vals.empty()) c.fine_grained_timing = true;
        else { const std::string v = vals[0]; c.fine_grained_timing = (v == "true" || v == "1" || v == "on" || v == "yes"); }
    }
    else if (key == "ct_dir") c.ct_dir = one();
    else if (key == "rtplan") c.rtplan = one();
    else if (key == "output_directory") c.output_directory = one();
    else if (key == "lut_dir") c.lut_dir = one();
    else if (key == "spot_list") c.spot_list = one();
    else if (key == "dump_spot_list") c.dump_spot_list = one();
    else if (key == "beams") { if (vals.empty()) usageError("--beams: At least 1 required"); c.beams = vals; }
    else if (key == "water_cube") {
        if (vals.empty()) c.water_cube = true;
        else { const std::string v = vals[0]; c.water_cube = (v == "true" || v == "1" || v == "on" || v == "yes"); }
    }
    else if (key == "water_cube_edge") c.water_cube_edge = (unsigned int)parseUInt(key, one(), 4096);
    else if (key == "layers") c.layers = (unsigned int)parseUInt(key, one(), 256);
    else if (key == "tracer_steps") c.tracer_steps = (unsigned int)parseUInt(key, one(), 4096);
    else if (key == "start_depth") { char* end = nullptr; c.start_depth = std::strtof(one().c_str(), &end); if (*end) usageError("--start_depth: not a number"); }
    else usageError("The following argument was not expected: --" + key);
    seen[key] = true;
}

void readConfigFile(Config& c, const std::string& path, std::map<std::string, bool>& seen) {
    std::ifstream in(path.c_str());
    if (!in) usageError("--config_file: File does not exist: " + path);
    std::string line;
    while (std::getline(in, line)) {
        const size_t hash = line.find_first_of("#;");
        if (hash != std::string::npos) line = line.substr(0, hash);
        line = trim(line);
        if (line.empty() || line.front() == '[') continue;            // blank, comment or [section]
        const size_t eq = line.find('=');
        if (eq == std::string::npos) usageError("--config_file: line without '=': " + line);
        std::string key = trim(line.substr(0, eq));
        while (!key.empty() && key.front() == '-') key.erase(key.begin());
        assign(c, key, splitValues(line.substr(eq + 1)), seen);
    }
}

std::string configToStr(const Config& c) {                            // like CLI11's config_to_str(default_also = true)
    std::ostringstream o;
    o << "gpu_id=" << c.gpu_id << "\n";
    o << "ct_dir=\"" << c.ct_dir << "\"\n";
    o << "rtplan=\"" << c.rtplan << "\"\n";
    o << "beams=[";
    for (size_t i = 0; i < c.beams.size(); ++i) o << (i ? ", " : "") << '"' << c.beams[i] << '"';
    o << "]\n";
    o << "output_directory=\"" << c.output_directory << "\"\n";
    o << "lut_dir=\"" << c.lut_dir << "\"\n";
    o << "water_cube=" << (c.water_cube ? "true" : "false") << "\n";
    o << "water_cube_edge=" << c.water_cube_edge << "\n";
    o << "layers=" << c.layers << "\n";
    o << "spot_list=\"" << c.spot_list << "\"\n";
    return o.str();
}

}  // namespace

int main(int argc, char** argv) {
    // pass 1: split the command line into (key, values); --config_file is applied first, the command line overrides it
    std::vector<std::pair<std::string, std::vector<std::string>>> cli;
    std::string configFile;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") { printHelp(); return 0; }
        if (a.rfind("--", 0) != 0) usageError("The following argument was not expected: " + a);
        std::string key = a.substr(2);
        std::vector<std::string> vals;
        const size_t eq = key.find('=');
        if (eq != std::string::npos) { vals.push_back(key.substr(eq + 1)); key = key.substr(0, eq); }
        while (i + 1 < argc && std::string(argv[i + 1]).rfind("--", 0) != 0) {
            vals.push_back(argv[++i]);
            if (key != "beams" && key != "gpu_ids") break;              // only --beams and --gpu_ids take several values
        }
        if (key == "config_file") {
            if (vals.size() != 1) usageError("--config_file: 1 required TEXT missing");
            configFile = vals[0];
        } else {
            cli.emplace_back(key, vals);
        }
    }
    Config config;
    std::map<std::string, bool> seen;
    if (!configFile.empty()) readConfigFile(config, configFile, seen);
    for (const auto& kv : cli) assign(config, kv.first, kv.second, seen);
    if (config.lut_dir.empty()) {
        const char* env = std::getenv("RTD_LUT_DIR");
        config.lut_dir = env ? env : "LUTs/";
    }
    if (!config.lut_dir.empty() && config.lut_dir.back() != '/') config.lut_dir += '/';

    // validation, in the reference's terms (config.cpp: ->required(), CLI::ExistingFile, CLI::ExistingDirectory)
    if (!seen.count("output_directory")) usageError("--output_directory is required");
    if (!isDir(config.output_directory)) usageError("--output_directory: Directory does not exist: " + config.output_directory);
    if (!config.water_cube) {
        if (!seen.count("ct_dir")) usageError("--ct_dir is required");
        if (!seen.count("rtplan")) usageError("--rtplan is required");
        if (!isFile(config.rtplan)) usageError("--rtplan: File does not exist: " + config.rtplan);
        if (!seen.count("beams")) usageError("--beams is required");
    }
    if (!isDir(config.lut_dir)) usageError("--lut_dir: Directory does not exist: " + config.lut_dir);
    if (!config.spot_list.empty() && !isFile(config.spot_list)) usageError("--spot_list: File does not exist: " + config.spot_list);
    if (!config.spot_list.empty() && !config.water_cube) usageError("--spot_list requires --water_cube (CT input is not built yet)");
    std::cout << configToStr(config) << std::endl;

    const std::vector<int> gpus = config.gpu_ids.empty() ? std::vector<int>{(int)config.gpu_id} : config.gpu_ids;
    rtd_options opt;
    rtd_default_options(&opt);
    opt.fine_grained_timing = config.fine_grained_timing ? 1 : 0;
    try {
        if (!config.water_cube) {                                       // DICOM input (SURVEY section 8 rows f2 + f3)
            runDicomPlan(config.lut_dir, config.output_directory, config.ct_dir, config.rtplan, config.beams, gpus,
                         config.start_depth, (int)config.tracer_steps, nullptr, &opt);
            return 0;
        }
        if (!config.dump_spot_list.empty()) {
            EnergyStruct ciddData = energyReader(config.lut_dir, /*waterCubeTest=*/true);
            writeSpotList(config.dump_spot_list, waterCubeSpots(ciddData, config.layers), rtd_plan::FieldGeometry());
            std::cout << "Written " << config.dump_spot_list << std::endl;
            return 0;
        }
        if (!config.spot_list.empty())
            runSpotListOnWaterCube(config.lut_dir, config.output_directory, config.water_cube_edge, config.spot_list, gpus, &opt);
        else
        runWaterCube(config.lut_dir, config.output_directory, config.water_cube_edge, config.layers, gpus, &opt);
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
