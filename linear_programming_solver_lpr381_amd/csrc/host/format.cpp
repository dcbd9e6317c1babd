// host/format.cpp -- the reference's text formats: .NET "0.###" / "F3" number formatting,
// AppendTableau (Models/PrimalSimplex.cs:272-304, Models/DualSimplex.cs:248-281) and
// AppendCanonicalForm (Models/PrimalSimplex.cs:259-270).  Host-side string work only.
#include "model.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace lpx { namespace host {

// Decimal rendering with `dec` fractional digits, midpoints away from zero (.NET custom and
// standard numeric format strings round the decimal digit string half away from zero).  A double
// sits exactly on a 10^-dec midpoint only when it is a dyadic rational, which "%.*f" (exact,
// round-half-even in glibc) would send the other way on even neighbours; nudge those.
static std::string fixed_away(double v, int dec)
{
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "\xE2\x88\x9E" : "-\xE2\x88\x9E";
    char buf[512];
    double a = std::fabs(v);
    bool tie = false;
    if (a < 4.0e15) {
        double scaled = a * std::pow(10.0, dec);
        // exact tie <=> scaled has fractional part exactly .5 AND the product above was exact;
        // verify with the exact decimal expansion instead of trusting the multiply
        (void)scaled;
        std::snprintf(buf, sizeof(buf), "%.*f", dec + 60, a);
        const char* dot = std::strchr(buf, '.');
        if (dot) {
            const char* p = dot + 1 + dec;
            if (*p == '5') { tie = true; for (const char* q = p + 1; *q; ++q) if (*q != '0') { tie = false; break; } }
        }
    }
    if (tie) {
        // round the magnitude up explicitly
        std::snprintf(buf, sizeof(buf), "%.*f", dec + 1, a);          // ...d5 exactly
        std::string s(buf);
        s.pop_back();                                                // drop the 5
        // increment the decimal string
        int i = (int)s.size() - 1;
        while (i >= 0) {
            if (s[i] == '.') { --i; continue; }
            if (s[i] == '9') { s[i] = '0'; --i; } else { s[i]++; break; }
        }
        if (i < 0) s.insert(s.begin(), '1');
        if (dec == 0 && !s.empty() && s.back() == '.') s.pop_back();
        return (std::signbit(v) ? "-" : "") + s;
    }
    std::snprintf(buf, sizeof(buf), "%.*f", dec, v);
    return buf;
}

// double.ToString("0.###"): up to three decimals, trailing zeros dropped.  .NET Core 3.0+ keeps the
// sign of values that round to zero ("-0").
std::string FormatNumber(double v)
{
    std::string s = fixed_away(v, 3);
    if (s.find('.') != std::string::npos) {
        while (!s.empty() && s.back() == '0') s.pop_back();
        if (!s.empty() && s.back() == '.') s.pop_back();
    }
    return s;
}

std::string FormatF(double v, int decimals) { return fixed_away(v, decimals); }

// Math.Round(v, decimals): round-half-to-even on the scaled value (as the BCL does for |v| < 1e16).
double RoundHalfEven(double v, int decimals)
{
    if (std::fabs(v) >= 1e16) return v;
    double p = std::pow(10.0, decimals);
    return std::nearbyint(v * p) / p;
}

std::string FormatRound3(double v) { return FormatNumber(RoundHalfEven(v, 3)); }

static std::string pad_left(const std::string& s, size_t w)
{
    return s.size() >= w ? s : std::string(w - s.size(), ' ') + s;
}

// AppendTableau: 12-character right-aligned columns, header, dash line, z row first.
std::string AppendTableau(const std::string& title, const double* T, int R, int C,
                          const std::vector<int32_t>& basis, const std::vector<std::string>& varNames, int iter)
{
    const int m = R - 1, ns = C - 1, w = 12;
    std::string sb;
    sb.reserve((size_t)(R + 3) * (C + 1) * w);
    sb += title + " " + std::to_string(iter) + "\n";
    sb += pad_left("Basis", w);
    for (int j = 0; j < ns; ++j) sb += pad_left(varNames[j], w);
    sb += pad_left("RHS", w);
    sb += "\n";
    sb += std::string((size_t)w * (ns + 2), '-') + "\n";
    sb += pad_left("z", w);
    for (int j = 0; j < ns; ++j) sb += pad_left(FormatNumber(T[(size_t)m * C + j]), w);
    sb += pad_left(FormatNumber(T[(size_t)m * C + ns]), w);
    sb += "\n";
    for (int i = 0; i < m; ++i) {
        sb += pad_left(varNames[basis[i]], w);
        for (int j = 0; j < ns; ++j) sb += pad_left(FormatNumber(T[(size_t)i * C + j]), w);
        sb += pad_left(FormatNumber(T[(size_t)i * C + ns]), w);
        sb += "\n";
    }
    return sb;
}

// MatrixToString, Models/RevisedPrimalSimplex.cs:248-261: 12-character right-aligned "0.###" cells.
std::string MatrixToString(const double* M, int r, int c)
{
    std::string sb;
    sb.reserve((size_t)r * ((size_t)c * 12 + 1));
    for (int i = 0; i < r; ++i) {
        for (int j = 0; j < c; ++j) sb += pad_left(FormatNumber(M[(size_t)i * c + j]), 12);
        sb += "\n";
    }
    return sb;
}

// BuildIterationBlock, Models/RevisedPrimalSimplex.cs:191-246.  rN / d may be null (iteration 0); entering < 0
// and a NaN theta stand for the reference's null arguments.  Nidx is the list AFTER the pivot while rN is
// in the order priced BEFORE it -- the reference labels them that way (:219-220) and so does this.
std::string BuildIterationBlock(int iter, const std::vector<int32_t>& Bidx, const std::vector<int32_t>& Nidx,
                                const std::vector<std::string>& names, const double* Binv, int m,
                                const std::vector<double>& xB, double z, const std::vector<double>* rN,
                                int entering, const std::vector<double>* d, double bestTheta, double eps)
{
    auto join = [](const std::vector<std::string>& v) {
        std::string o; for (size_t i = 0; i < v.size(); ++i) { if (i) o += ", "; o += v[i]; } return o; };
    std::string sb = "=== Revised Simplex Iteration " + std::to_string(iter) + " ===\n";
    std::vector<std::string> t;
    for (int32_t k : Bidx) t.push_back(names[k]);
    sb += "Basis: " + join(t) + "\n";
    t.clear(); for (int32_t k : Nidx) t.push_back(names[k]);
    sb += "Nonbasic: " + join(t) + "\n";
    sb += "\nProduct-form: current B^{-1}\n";
    sb += MatrixToString(Binv, m, m);
    t.clear(); for (double v : xB) t.push_back(FormatNumber(v));
    sb += "x_B = [" + join(t) + "]\n";
    sb += "z = " + FormatNumber(z) + "\n";
    if (rN) {
        sb += "\nReduced costs (r_N = c_N - c_B^T B^{-1} N):\n";
        for (size_t j = 0; j < rN->size(); ++j)
            sb += "  r(" + std::to_string(Nidx[j]) + ":" + names[Nidx[j]] + ") = " + FormatNumber((*rN)[j]) + "\n";
    }
    if (entering >= 0) sb += "\nEntering variable: " + names[entering] + "\n";
    if (d) {
        sb += "Direction d = B^{-1} * a_entering:\n";
        t.clear(); for (double v : *d) t.push_back(FormatNumber(v));
        sb += "  d = [" + join(t) + "]\n";
        sb += "\nRatio test (theta):\n";
        for (size_t i = 0; i < d->size(); ++i) {
            if ((*d)[i] > eps)
                sb += "  row " + std::to_string(i + 1) + ": " + FormatNumber(xB[i]) + " / " + FormatNumber((*d)[i]) +
                      " = " + FormatNumber(xB[i] / (*d)[i]) + "\n";
            else
                sb += "  row " + std::to_string(i + 1) + ": d_i <= 0 (skip)\n";
        }
        if (!std::isnan(bestTheta)) sb += "Chosen theta* = " + FormatNumber(bestTheta) + "\n";
    }
    sb += "\n";
    return sb;
}

static std::string term(double v, int j)
{
    return std::string(v >= 0 ? "+" : "-") + FormatNumber(std::fabs(v)) + "x" + std::to_string(j + 1);
}

// AppendCanonicalForm, Models/PrimalSimplex.cs:259-270
std::string AppendCanonicalForm(const LPProblem& model)
{
    std::string sb = "Objective: max ";
    for (size_t j = 0; j < model.C.size(); ++j) { if (j) sb += " "; sb += term(model.C[j], (int)j); }
    sb += "\nSubject to:\n";
    for (const Constraint& c : model.Constraints) {
        sb += "  ";
        for (size_t j = 0; j < c.A.size(); ++j) { if (j) sb += " "; sb += term(c.A[j], (int)j); }
        const char* rel = c.Relation == Rel::LE ? "<=" : (c.Relation == Rel::GE ? ">=" : "=");
        sb += std::string(" ") + rel + " " + FormatNumber(c.B) + "\n";
    }
    sb += "x >= 0\n";
    return sb;
}

}}  // namespace lpx::host
