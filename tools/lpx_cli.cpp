// lpx_cli -- Linux stand-in for the reference's WinForms host (Form1.cs), over the C ABI of liblpx.so only.
//
//   lpx_cli [--algorithm NAME] [--repaired] [--iterations] [--export FILE] INPUT.txt
//
// Does what Form1 does around the solvers: reads the model text (Import, Form1.cs:284-296), parses it with the LPParser
// grammar (lpx_parse_text, Models/LPParser.cs:9-79), runs the algorithm chosen by its dropdown name (btnSolve_Click,
// Form1.cs:231-279), shows the iteration text followed by "Final Report:" and "Summary:" (:277-278), and can write the
// export file layout of BtnExport_Click (:308-315).  C only touches include/lpx.h: this is also the link test of the
// boundary from a compiled host.  There is no CPU fallback: without a gfx950 device the solve fails with LPX_EDEVICE.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

#include "../include/lpx.h"

static std::string g_iterations;
static void on_text(void*, const char* text, const uint8_t*, int, int) { g_iterations += text; }

int main(int argc, char** argv)
{
    std::string algorithm = "Primal Simplex", input, exportPath;
    bool repaired = false, iterations = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--algorithm" && i + 1 < argc) algorithm = argv[++i];
        else if (a == "--repaired") repaired = true;
        else if (a == "--iterations") iterations = true;
        else if (a == "--export" && i + 1 < argc) exportPath = argv[++i];
        else if (a == "--help" || a == "-h") {
            std::printf("usage: lpx_cli [--algorithm NAME] [--repaired] [--iterations] [--export FILE] INPUT.txt\n"
                        "  NAME: Primal Simplex | Revised Primal Simplex | Dual Simplex | Branch and Bound |\n"
                        "        Revised Branch and Bound | Branch and Bound Knapsack | Cutting Plane | Revised Cutting Plane\n");
            return 0;
        } else input = a;
    }
    if (input.empty()) { std::fprintf(stderr, "lpx_cli: no input file (try --help)\n"); return 64; }
    std::ifstream f(input);
    if (!f) { std::fprintf(stderr, "Error reading file: %s\n", input.c_str()); return 66; }
    std::stringstream ss; ss << f.rdbuf();
    const std::string text = ss.str();

    lpx_parsed p;
    char err[1024];
    if (lpx_parse_text(text.c_str(), &p) != 0) { lpx_last_error(err, sizeof err); std::fprintf(stderr, "%s\n", err); return 65; }
    lpx_problem prob; prob.sense = p.sense; prob.n = p.n; prob.m = p.m; prob.c = p.c; prob.A = p.A; prob.rel = p.rel; prob.b = p.b;
    lpx_solve_opts o; lpx_default_solve_opts(&o);
    o.text_cb = on_text;
    o.render_iterations = iterations ? 1 : 0;
    if (repaired) { o.dual_flags = 7; o.bnb_mode = 1; }
    lpx_result r;
    const int rc = lpx_solve(&prob, algorithm.c_str(), &o, &r);
    lpx_parsed_free(&p);
    if (rc != 0) { lpx_last_error(err, sizeof err); std::fprintf(stderr, "%s\n", err); return rc == LPX_EDEVICE ? 69 : 70; }
    std::string shown = g_iterations;
    shown += "\n\nFinal Report:\n"; shown += r.report ? r.report : "";
    shown += "\n\nSummary:\n"; shown += r.summary ? r.summary : "";
    std::fputs(shown.c_str(), stdout); std::fputc('\n', stdout);
    if (!exportPath.empty()) {
        std::ofstream w(exportPath);
        w << "Linear Program:\n" << text << "\n\nIterations:\n" << shown << "\n";
    }
    lpx_result_free(&r);
    return 0;
}
