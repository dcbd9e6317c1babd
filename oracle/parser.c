/*
 * oracle/parser.c -- CPU restatement of Models/LPParser.cs:9-79 (TEST INFRASTRUCTURE, see
 * lpx_oracle.h).  Grammar: line 1 `^(max|min)\s*:\s*(.+)$` (case-insensitive, :19); every other
 * non-blank line `^(.+?)(<=|>=|=)(.+)$` (:36); terms after "-" -> "+-" and blank removal, split
 * on '+', each `^([-]?\d*\.?\d*)x\d+$` (:63-69); empty coefficient = 1, "-" = -1 (:73-74).
 * The digits after `x` are ignored: coefficients are POSITIONAL (:66-76).
 */
#include "lpx_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <ctype.h>

static void seterr(char* err, int errlen, const char* msg, const char* arg)
{
    if (err && errlen > 0) snprintf(err, errlen, "%s%s", msg, arg ? arg : "");
}

static char* trimdup(const char* s, size_t len)
{
    while (len > 0 && isspace((unsigned char)*s)) { s++; len--; }
    while (len > 0 && isspace((unsigned char)s[len - 1])) len--;
    char* r = (char*)malloc(len + 1);
    memcpy(r, s, len); r[len] = 0;
    return r;
}

/* ParseCoefficients, :61-79.  Returns count or -1. */
static int parse_coeffs(const char* expr, double** out, char* err, int errlen)
{
    size_t L = strlen(expr);
    char* buf = (char*)malloc(2 * L + 2);
    size_t k = 0;
    for (size_t i = 0; i < L; i++) {           /* Replace("-", "+-").Replace(" ", "") */
        if (expr[i] == '-') { buf[k++] = '+'; buf[k++] = '-'; }
        else if (expr[i] == ' ') continue;
        else buf[k++] = expr[i];
    }
    buf[k] = 0;
    int cap = 16, cnt = 0;
    double* v = (double*)malloc(sizeof(double) * cap);
    char* s = buf;
    while (*s) {
        char* e = strchr(s, '+');
        size_t len = e ? (size_t)(e - s) : strlen(s);
        if (len > 0) {                         /* RemoveEmptyEntries */
            char* part = trimdup(s, len);
            /* ^([-]?\d*\.?\d*)x\d+$ */
            size_t i = 0, pl = strlen(part);
            if (part[i] == '-') i++;
            while (isdigit((unsigned char)part[i])) i++;
            if (part[i] == '.') i++;
            while (isdigit((unsigned char)part[i])) i++;
            size_t vend = i;
            int ok = (part[i] == 'x');
            if (ok) { i++; size_t d0 = i; while (isdigit((unsigned char)part[i])) i++; ok = (i > d0) && (i == pl); }
            if (!ok) { seterr(err, errlen, "Cannot parse coefficient: ", part); free(part); free(v); free(buf); return -1; }
            double val;
            if (vend == 0) val = 1;                                   /* :73 */
            else if (vend == 1 && part[0] == '-') val = -1;           /* :74 */
            else {
                char tmp[64]; size_t tl = vend < 63 ? vend : 63;
                memcpy(tmp, part, tl); tmp[tl] = 0;
                int hasdigit = 0; for (size_t q = 0; q < tl; q++) if (isdigit((unsigned char)tmp[q])) hasdigit = 1;
                if (!hasdigit) { seterr(err, errlen, "Cannot parse coefficient: ", part); free(part); free(v); free(buf); return -1; }
                val = strtod(tmp, NULL);                              /* double.Parse */
            }
            if (cnt == cap) { cap *= 2; v = (double*)realloc(v, sizeof(double) * cap); }
            v[cnt++] = val;
            free(part);
        }
        if (!e) break;
        s = e + 1;
    }
    free(buf);
    *out = v;
    return cnt;
}

void orc_parsed_free(orc_parsed* p)
{
    if (!p) return;
    free(p->c); free(p->A); free(p->rel); free(p->b);
    memset(p, 0, sizeof(*p));
}

int orc_parse_text(const char* text, orc_parsed* out, char* err, int errlen)
{
    memset(out, 0, sizeof(*out));
    /* split on \r \n, trim, drop blank (:11-14) */
    int nl = 0, capl = 16;
    char** lines = (char**)malloc(sizeof(char*) * capl);
    const char* s = text;
    while (*s) {
        const char* e = s;
        while (*e && *e != '\r' && *e != '\n') e++;
        char* t = trimdup(s, (size_t)(e - s));
        if (*t) { if (nl == capl) { capl *= 2; lines = (char**)realloc(lines, sizeof(char*) * capl); } lines[nl++] = t; }
        else free(t);
        s = *e ? e + 1 : e;
    }
    int rc = 0;
    double** rows = NULL; int* rowlen = NULL;
    if (nl < 2) { seterr(err, errlen, "Input must contain an objective and at least one constraint.", NULL); rc = -1; goto out; }
    {
        /* ^(max|min)\s*:\s*(.+)$  IgnoreCase (:19) */
        const char* l0 = lines[0];
        int sense;
        if (strncasecmp(l0, "max", 3) == 0) sense = ORC_MAX;
        else if (strncasecmp(l0, "min", 3) == 0) sense = ORC_MIN;
        else { seterr(err, errlen, "Objective format incorrect. Example: Max: 3x1 + 5x2", NULL); rc = -2; goto out; }
        const char* q = l0 + 3;
        while (isspace((unsigned char)*q)) q++;
        if (*q != ':') { seterr(err, errlen, "Objective format incorrect. Example: Max: 3x1 + 5x2", NULL); rc = -2; goto out; }
        q++;
        while (isspace((unsigned char)*q)) q++;
        if (!*q) { seterr(err, errlen, "Objective format incorrect. Example: Max: 3x1 + 5x2", NULL); rc = -2; goto out; }
        double* c = NULL;
        int n = parse_coeffs(q, &c, err, errlen);
        if (n < 0) { rc = -3; goto out; }
        out->sense = sense; out->n = n; out->c = c;
    }
    {
        int m = nl - 1;
        out->m = m;
        out->rel = (int32_t*)malloc(sizeof(int32_t) * m);
        out->b = (double*)malloc(sizeof(double) * m);
        rows = (double**)calloc(m, sizeof(double*));
        rowlen = (int*)calloc(m, sizeof(int));
        for (int i = 0; i < m; i++) {
            const char* l = lines[i + 1];
            size_t L = strlen(l);
            /* ^(.+?)(<=|>=|=)(.+)$ : smallest non-empty LHS followed by a relation and a non-empty RHS (:36) */
            size_t pos = 0; int rel = -1; size_t rl = 0;
            for (size_t k = 1; k < L; k++) {
                if (l[k] == '<' && l[k + 1] == '=' && k + 2 < L) { pos = k; rel = ORC_LE; rl = 2; break; }
                if (l[k] == '>' && l[k + 1] == '=' && k + 2 < L) { pos = k; rel = ORC_GE; rl = 2; break; }
                if (l[k] == '=' && k + 1 < L) { pos = k; rel = ORC_EQ; rl = 1; break; }
            }
            if (rel < 0) { seterr(err, errlen, "Constraint format incorrect: ", l); rc = -4; goto out; }
            char* lhs = trimdup(l, pos);
            char* rhs = trimdup(l + pos + rl, L - pos - rl);
            int cnt = parse_coeffs(lhs, &rows[i], err, errlen);
            free(lhs);
            if (cnt < 0) { free(rhs); rc = -3; goto out; }
            rowlen[i] = cnt;
            char* endp = NULL;
            double B = strtod(rhs, &endp);                            /* double.TryParse (:53) */
            while (endp && isspace((unsigned char)*endp)) endp++;
            if (endp == rhs || (endp && *endp)) { seterr(err, errlen, "Invalid RHS number: ", rhs); free(rhs); rc = -5; goto out; }
            free(rhs);
            out->rel[i] = rel; out->b[i] = B;
            if (cnt < out->n) out->ragged = 1;   /* later IndexOutOfRange in BuildTableau (PrimalSimplex.cs:190) */
        }
        out->A = (double*)calloc((size_t)m * (out->n > 0 ? out->n : 1), sizeof(double));
        for (int i = 0; i < m; i++)
            for (int j = 0; j < out->n && j < rowlen[i]; j++) out->A[(size_t)i * out->n + j] = rows[i][j];
    }
out:
    if (rows) { for (int i = 0; i < out->m; i++) free(rows[i]); free(rows); }
    free(rowlen);
    for (int i = 0; i < nl; i++) free(lines[i]);
    free(lines);
    if (rc) orc_parsed_free(out);
    return rc;
}
