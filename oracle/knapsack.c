/*
 * oracle/knapsack.c -- CPU restatement of Models/BranchAndBoundKnapsack.cs (TEST
 * INFRASTRUCTURE, see lpx_oracle.h).  Best-first B&B for a single-constraint 0/1 knapsack with
 * the greedy fractional bound; the same array heap (Push sift-up breaks on `<= 0`, Pop swaps
 * last and sifts down, :494-547) so that the pop order among equal bounds is reproduced.
 *
 * Nodes store only their fixed decisions (index,value) instead of the reference's int[n]
 * `Assigned` (:25); the assignment vector is materialised into a scratch array before each
 * ComputeRelaxation, which leaves every arithmetic step and its order unchanged.  The O(n)
 * `_itemsByRatio.First(x => x.Index == i)` lookup (:447) is replaced by direct indexing (same
 * item, same values).  Report text (:139-143 etc.) is not produced.
 */
#include "lpx_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define KEPS 1e-9   /* :56 */

typedef struct {
    int depth;
    int32_t* idx;      /* fixed original indices, in decision order */
    int8_t* val;       /* 0/1 */
    double bound, profit, weight;
} knode;

typedef struct {
    int n; double cap;
    const double* profit; const double* weight;
    int32_t* order;    /* ratio order -> original index */
    int8_t* assigned;  /* scratch: -1/0/1 */
    double* relaxed;   /* scratch [n] */
    orc_knap_result* out;
} kctx;

void orc_knap_result_free(orc_knap_result* r)
{
    if (!r) return;
    free(r->best_x);
    memset(r, 0, sizeof(*r));
}

/* Item.Ratio, :19 */
static double ratio_of(double p, double w) { return w > 0 ? p / w : INFINITY; }

typedef struct { double ratio, profit; int32_t idx; } sitem;
static int cmp_items(const void* a, const void* b)
{
    const sitem* x = (const sitem*)a; const sitem* y = (const sitem*)b;
    /* OrderByDescending(Ratio).ThenByDescending(Profit), stable (:75-79) */
    if (x->ratio > y->ratio) return -1;
    if (x->ratio < y->ratio) return 1;
    if (x->profit > y->profit) return -1;
    if (x->profit < y->profit) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

void orc_knapsack_order(const double* profit, const double* weight, int n, int32_t* order)
{
    sitem* it = (sitem*)malloc(sizeof(sitem) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) { it[i].ratio = ratio_of(profit[i], weight[i]); it[i].profit = profit[i]; it[i].idx = i; }
    qsort(it, n, sizeof(sitem), cmp_items);
    for (int i = 0; i < n; i++) order[i] = it[i].idx;
    free(it);
}

/* ComputeRelaxation, :431-491.  assigned8: -1/0/1 per ORIGINAL index. */
static void relax8(const double* profit, const double* weight, int n, double cap,
                   const int32_t* order, const int8_t* assigned, double* relaxed,
                   double* out_profit, double* out_weight, int32_t* out_frac)
{
    double w = 0.0, p = 0.0;
    int32_t frac = -1;
    if (relaxed) for (int i = 0; i < n; i++) relaxed[i] = 0.0;
    for (int i = 0; i < n; i++) {                                     /* :442-452 */
        if (assigned[i] == 1) {
            if (relaxed) relaxed[i] = 1.0;
            w += weight[i];
            p += profit[i];
        }
    }
    if (w > cap + KEPS) { *out_profit = p; *out_weight = w; *out_frac = -1; return; }   /* :455-456 */
    for (int s = 0; s < n; s++) {                                     /* :459-488 */
        int orig = order[s];
        if (assigned[orig] == 1) continue;
        if (assigned[orig] == 0) continue;
        double wi = weight[orig];
        if (w + wi <= cap + KEPS) {
            if (relaxed) relaxed[orig] = 1.0;
            w += wi;
            p += profit[orig];
        } else {
            double remain = cap - w;
            if (remain > KEPS && wi > KEPS) {
                double fr = remain / wi;
                if (relaxed) relaxed[orig] = fr;
                p += profit[orig] * fr;
                w += wi * fr;
                frac = s;
            }
            break;
        }
    }
    *out_profit = p; *out_weight = w; *out_frac = frac;
}

void orc_knapsack_relax(const double* profit, const double* weight, int n, double cap,
                        const int32_t* order, const int32_t* assigned,
                        double* relaxed, double* out_profit, double* out_weight,
                        int32_t* out_frac_sorted_idx)
{
    int8_t* a8 = (int8_t*)malloc(n > 0 ? n : 1);
    for (int i = 0; i < n; i++) a8[i] = (int8_t)assigned[i];
    relax8(profit, weight, n, cap, order, a8, relaxed, out_profit, out_weight, out_frac_sorted_idx);
    free(a8);
}

/* ---- SimpleMaxHeap<Node>, :494-547 (cmp = Bound.CompareTo) -------------------------------- */
typedef struct { knode** d; int64_t n, cap; } kheap;
static int cmpb(const knode* a, const knode* b) { return (a->bound > b->bound) - (a->bound < b->bound); }
static void hswap(kheap* h, int64_t a, int64_t b) { knode* t = h->d[a]; h->d[a] = h->d[b]; h->d[b] = t; }
static void hpush(kheap* h, knode* x)
{
    if (h->n == h->cap) { h->cap = h->cap ? h->cap * 2 : 1024; h->d = (knode**)realloc(h->d, sizeof(knode*) * h->cap); }
    h->d[h->n++] = x;
    int64_t ci = h->n - 1;
    while (ci > 0) {
        int64_t pi = (ci - 1) / 2;
        if (cmpb(h->d[ci], h->d[pi]) <= 0) break;
        hswap(h, ci, pi);
        ci = pi;
    }
}
static knode* hpop(kheap* h)
{
    int64_t li = h->n - 1;
    hswap(h, 0, li);
    knode* ret = h->d[li];
    h->n--;
    li = h->n - 1;
    int64_t i = 0;
    for (;;) {
        int64_t l = 2 * i + 1, r = 2 * i + 2, largest = i;
        if (l <= li && cmpb(h->d[l], h->d[largest]) > 0) largest = l;
        if (r <= li && cmpb(h->d[r], h->d[largest]) > 0) largest = r;
        if (largest == i) break;
        hswap(h, i, largest);
        i = largest;
    }
    return ret;
}

static void node_free(knode* k) { if (k) { free(k->idx); free(k->val); free(k); } }
static void materialise(kctx* c, const knode* k, int on)
{
    for (int t = 0; t < k->depth; t++) c->assigned[k->idx[t]] = on ? k->val[t] : (int8_t)-1;
}
static knode* child_of(const knode* k, int item, int v)
{
    knode* ch = (knode*)calloc(1, sizeof(knode));
    ch->depth = k->depth + 1;
    ch->idx = (int32_t*)malloc(sizeof(int32_t) * ch->depth);
    ch->val = (int8_t*)malloc(ch->depth);
    memcpy(ch->idx, k->idx, sizeof(int32_t) * k->depth);
    memcpy(ch->val, k->val, k->depth);
    ch->idx[k->depth] = item; ch->val[k->depth] = (int8_t)v;
    return ch;
}
static void do_relax(kctx* c, int want_vec, double* p, double* w, int32_t* f)
{
    c->out->relaxations++;
    relax8(c->profit, c->weight, c->n, c->cap, c->order, c->assigned, want_vec ? c->relaxed : NULL, p, w, f);
}
/* relaxed.All(v => |v - Math.Round(v)| < EPS), :225/:287 -- only the fractional entry can fail */
static int all_int(kctx* c, int32_t frac, double weight_before_frac_unused)
{
    (void)weight_before_frac_unused;
    if (frac < 0) return 1;
    double v = c->relaxed[c->order[frac]];
    return fabs(v - rint(v)) < KEPS;
}

/* one child of :207-264 / :267-327 */
static void eval_child(kctx* c, kheap* pq, const knode* node, int origIdx, int v)
{
    knode* ch = child_of(node, origIdx, v);
    c->assigned[origIdx] = (int8_t)v;
    double p, w; int32_t f;
    do_relax(c, 1, &p, &w, &f);
    orc_knap_result* o = c->out;
    if (w > c->cap + KEPS) {                                           /* :215 / :277 */
        node_free(ch);
    } else if (p > o->best_z + KEPS) {                                 /* :223 / :285 */
        int allInt = all_int(c, f, 0);
        int feasible = w <= c->cap + KEPS;
        if (allInt && feasible) {                                      /* :228-236 */
            if (p > o->best_z + KEPS) {
                o->best_z = p; o->status = 0;
                for (int i = 0; i < c->n; i++) o->best_x[i] = (int32_t)rint(c->relaxed[i]);
            }
            node_free(ch);
        } else {                                                       /* :239-248 */
            ch->bound = p; ch->profit = p; ch->weight = w;
            hpush(pq, ch);
            if (pq->n > o->max_heap) o->max_heap = pq->n;
        }
    } else {
        node_free(ch);                                                 /* :257-264 */
    }
    c->assigned[origIdx] = -1;
}

/* BranchAndBoundKnapsack.Solve, :58-407 */
int orc_knapsack_solve(const orc_problem* p, int64_t max_nodes, orc_knap_result* out)
{
    memset(out, 0, sizeof(*out));
    if (p->m != 1) return ORC_E_KNAP_SHAPE;                            /* :66-67 */
    if (p->rel[0] != ORC_LE) return ORC_E_KNAP_SHAPE;                  /* :69 */
    kctx c; memset(&c, 0, sizeof(c));
    int n = p->n;
    c.n = n; c.cap = p->b[0]; c.profit = p->c; c.weight = p->A; c.out = out;
    c.order = (int32_t*)malloc(sizeof(int32_t) * (n > 0 ? n : 1));
    c.assigned = (int8_t*)malloc(n > 0 ? n : 1);
    c.relaxed = (double*)malloc(sizeof(double) * (n > 0 ? n : 1));
    memset(c.assigned, -1, n);
    orc_knapsack_order(c.profit, c.weight, n, c.order);               /* :75-79 */
    out->n = n;
    out->best_x = (int32_t*)calloc(n > 0 ? n : 1, sizeof(int32_t));
    out->best_z = -INFINITY; out->status = 1;                          /* :82-83 */

    kheap pq; memset(&pq, 0, sizeof(pq));
    knode* root = (knode*)calloc(1, sizeof(knode));                    /* :102-113 */
    { double pr, w; int32_t f; do_relax(&c, 0, &pr, &w, &f); root->bound = pr; root->profit = pr; root->weight = w; }
    hpush(&pq, root);
    out->max_heap = 1;

    while (pq.n > 0) {                                                 /* :118 */
        if (max_nodes > 0 && out->nodes_popped >= max_nodes) break;
        knode* node = hpop(&pq);
        out->nodes_popped++;                                           /* :121 */
        if (node->bound <= out->best_z + KEPS) { node_free(node); continue; }   /* :124 */
        out->nodes_expanded++;
        materialise(&c, node, 1);
        double profitRel, weightRel; int32_t frac;
        do_relax(&c, 1, &profitRel, &weightRel, &frac);                /* :127 */
        if (frac == -1) {                                              /* :147-177 */
            if (weightRel <= c.cap + KEPS) {
                double cand = profitRel;
                if (cand > out->best_z + KEPS) {
                    out->best_z = cand; out->status = 0;
                    for (int i = 0; i < n; i++) out->best_x[i] = c.relaxed[i] >= 0.5 ? 1 : 0;
                }
            }
            materialise(&c, node, 0);
            node_free(node);
            continue;
        }
        int origIdx = c.order[frac];                                   /* :180 */
        eval_child(&c, &pq, node, origIdx, 0);                         /* LEFT  x=0, :207-264 */
        eval_child(&c, &pq, node, origIdx, 1);                         /* RIGHT x=1, :267-327 */
        materialise(&c, node, 0);
        node_free(node);
    }
    while (pq.n > 0) node_free(pq.d[--pq.n]);
    free(pq.d);
    free(c.order); free(c.assigned); free(c.relaxed);
    return 0;
}
