// host/parser.cpp -- LPParser.ParseFromText (Models/LPParser.cs:9-79): the reference's input file
// grammar.  Host-side; errors carry the reference's messages.
#include "model.h"

#include <cctype>
#include <cstdlib>
#include <cstring>

namespace lpx { namespace host {

static std::string trim(const std::string& s)
{
    size_t a = 0, b = s.size();
    while (a < b && std::isspace((unsigned char)s[a])) ++a;
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

// ParseCoefficients, :61-79: "-" -> "+-", blanks removed, split on '+', each term
// ^([-]?\d*\.?\d*)x\d+$ ; empty coefficient = 1, "-" = -1.  The subscript is ignored (positional).
static std::vector<double> ParseCoefficients(const std::string& expr0)
{
    std::string expr;
    for (char ch : expr0) {
        if (ch == '-') expr += "+-";
        else if (ch == ' ') continue;
        else expr += ch;
    }
    std::vector<double> out;
    size_t pos = 0;
    while (pos <= expr.size()) {
        size_t e = expr.find('+', pos);
        std::string part = trim(expr.substr(pos, e == std::string::npos ? std::string::npos : e - pos));
        if (!part.empty()) {
            size_t i = 0;
            if (i < part.size() && part[i] == '-') ++i;
            while (i < part.size() && std::isdigit((unsigned char)part[i])) ++i;
            if (i < part.size() && part[i] == '.') ++i;
            while (i < part.size() && std::isdigit((unsigned char)part[i])) ++i;
            const size_t vend = i;
            bool ok = i < part.size() && part[i] == 'x';
            if (ok) { ++i; size_t d0 = i; while (i < part.size() && std::isdigit((unsigned char)part[i])) ++i; ok = i > d0 && i == part.size(); }
            if (!ok) throw LpxException(LPX_E_PARSE, "Cannot parse coefficient: " + part);
            const std::string val = part.substr(0, vend);
            double v;
            if (val.empty()) v = 1;
            else if (val == "-") v = -1;
            else {
                bool digit = false; for (char ch : val) if (std::isdigit((unsigned char)ch)) digit = true;
                if (!digit) throw LpxException(LPX_E_PARSE, "Cannot parse coefficient: " + part);   // double.Parse(".") throws
                v = std::strtod(val.c_str(), nullptr);
            }
            out.push_back(v);
        }
        if (e == std::string::npos) break;
        pos = e + 1;
    }
    return out;
}

LPProblem ParseFromText(const std::string& input)
{
    std::vector<std::string> lines;
    size_t pos = 0;
    while (pos <= input.size()) {
        size_t e = input.find_first_of("\r\n", pos);
        std::string l = trim(input.substr(pos, e == std::string::npos ? std::string::npos : e - pos));
        if (!l.empty()) lines.push_back(l);
        if (e == std::string::npos) break;
        pos = e + 1;
    }
    if (lines.size() < 2) throw LpxException(LPX_E_PARSE, "Input must contain an objective and at least one constraint.");

    // ^(max|min)\s*:\s*(.+)$, IgnoreCase (:19)
    const std::string& l0 = lines[0];
    const char* bad_obj = "Objective format incorrect. Example: Max: 3x1 + 5x2";
    if (l0.size() < 3) throw LpxException(LPX_E_PARSE, bad_obj);
    std::string head = l0.substr(0, 3);
    for (char& ch : head) ch = (char)std::tolower((unsigned char)ch);
    Sense sense;
    if (head == "max") sense = Sense::Max; else if (head == "min") sense = Sense::Min;
    else throw LpxException(LPX_E_PARSE, bad_obj);
    size_t i = 3;
    while (i < l0.size() && std::isspace((unsigned char)l0[i])) ++i;
    if (i >= l0.size() || l0[i] != ':') throw LpxException(LPX_E_PARSE, bad_obj);
    ++i;
    while (i < l0.size() && std::isspace((unsigned char)l0[i])) ++i;
    if (i >= l0.size()) throw LpxException(LPX_E_PARSE, bad_obj);

    LPProblem problem;
    problem.ObjectiveSense = sense;
    problem.C = ParseCoefficients(l0.substr(i));

    for (size_t li = 1; li < lines.size(); ++li) {
        const std::string& l = lines[li];
        // ^(.+?)(<=|>=|=)(.+)$ : shortest non-empty LHS (:36)
        size_t at = std::string::npos, rl = 0; Rel rel = Rel::LE;
        for (size_t k = 1; k < l.size(); ++k) {
            if (l[k] == '<' && k + 1 < l.size() && l[k + 1] == '=' && k + 2 < l.size()) { at = k; rl = 2; rel = Rel::LE; break; }
            if (l[k] == '>' && k + 1 < l.size() && l[k + 1] == '=' && k + 2 < l.size()) { at = k; rl = 2; rel = Rel::GE; break; }
            if (l[k] == '=' && k + 1 < l.size()) { at = k; rl = 1; rel = Rel::EQ; break; }
        }
        if (at == std::string::npos) throw LpxException(LPX_E_PARSE, "Constraint format incorrect: " + l);
        std::string lhs = trim(l.substr(0, at));
        std::string rhs = trim(l.substr(at + rl));
        Constraint c;
        c.A = ParseCoefficients(lhs);
        c.Relation = rel;
        char* endp = nullptr;
        double B = std::strtod(rhs.c_str(), &endp);                       // double.TryParse (:53)
        while (endp && std::isspace((unsigned char)*endp)) ++endp;
        if (endp == rhs.c_str() || (endp && *endp)) throw LpxException(LPX_E_PARSE, "Invalid RHS number: " + rhs);
        c.B = B;
        problem.Constraints.push_back(c);
    }
    return problem;
}

}}  // namespace lpx::host
