/*
 * ca_oracle.cpp -- CPU ORACLE (test infrastructure, see ca_oracle.h for the parity status).
 *
 * Serial, one arena at a time, written for clarity: std::vector neighbour lists, plain loops.
 * Compile with -O2 -ffp-contract=off (no FMA contraction) so that every fp32 expression rounds
 * exactly like the HIP kernels, which are built with the same contract.
 *
 * Reference citations are into /root/reference/collision_avoidance/ :
 *   env.py  = envs/collision_avoidence_env.py      utils.py = envs/utils.py
 *   ALAN    = ALAN/ALAN_true.py
 * "App. A.x" = SURVEY.md Appendix A (behaviour contract of the external rvo2 module).
 */
#include "ca_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>
#include <utility>
#include <vector>

namespace {

/* ------------------------------------------------------------------------------------------ */
/* fp32 2-vectors with the operation order of the RVO2 library's Vector2                       */
/* ------------------------------------------------------------------------------------------ */
struct V2 {
    float x, y;
};
inline V2 mk(float x, float y) { return V2{x, y}; }
inline V2 operator+(V2 a, V2 b) { return mk(a.x + b.x, a.y + b.y); }
inline V2 operator-(V2 a, V2 b) { return mk(a.x - b.x, a.y - b.y); }
inline V2 operator-(V2 a) { return mk(-a.x, -a.y); }
inline V2 operator*(float s, V2 a) { return mk(s * a.x, s * a.y); }
inline V2 operator*(V2 a, float s) { return mk(a.x * s, a.y * s); }
inline float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
inline float det(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
inline float absSq(V2 a) { return dot(a, a); }
inline float vabs(V2 a) { return std::sqrt(absSq(a)); }
/* vector / scalar multiplies by the reciprocal (upstream Vector2::operator/) */
inline V2 vdiv(V2 a, float s) {
    const float inv = 1.0f / s;
    return mk(a.x * inv, a.y * inv);
}
inline V2 normalize(V2 a) { return vdiv(a, vabs(a)); }
inline float sqr(float a) { return a * a; }

const float EPS = 0.00001f; /* RVO_EPSILON, App. A */
const float FINF = std::numeric_limits<float>::infinity();

struct Line {
    V2 point, dir;
};

struct ObstVertex {
    V2 p, unitDir;
    int next, prev;
    bool convex;
};

struct AgentParams {
    float neighborDist;
    int maxNeighbors;
    float timeHorizon, timeHorizonObst, radius, maxSpeed;
};

/* leftOf(a,b,c) = det(a - c, b - a): >0 when c is left of the line a->b (App. A.2) */
inline float leftOf(V2 a, V2 b, V2 c) { return det(a - c, b - a); }

/* squared distance from c to segment ab (App. A.2 "segment distance") */
inline float distSqPointSegment(V2 a, V2 b, V2 c) {
    const float r = dot(c - a, b - a) / absSq(b - a);
    if (r < 0.0f) return absSq(c - a);
    if (r > 1.0f) return absSq(c - b);
    return absSq(c - (a + r * (b - a)));
}

/* append a closed polygon (>=2 vertices) to the vertex table: App. A.2 "Obstacle vertex
 * attributes at addObstacle" (what env.py:145 / ALAN:476 feed to sim.addObstacle) */
int add_polygon(std::vector<ObstVertex>& tab, const float* xy, int n) {
    if (n < 2) return -1;
    const int base = (int)tab.size();
    for (int i = 0; i < n; ++i) {
        ObstVertex v;
        v.p = mk(xy[2 * i], xy[2 * i + 1]);
        v.next = base + (i == n - 1 ? 0 : i + 1);
        v.prev = base + (i == 0 ? n - 1 : i - 1);
        const V2 pn = mk(xy[2 * (i == n - 1 ? 0 : i + 1)], xy[2 * (i == n - 1 ? 0 : i + 1) + 1]);
        const V2 pp = mk(xy[2 * (i == 0 ? n - 1 : i - 1)], xy[2 * (i == 0 ? n - 1 : i - 1) + 1]);
        v.unitDir = normalize(pn - v.p);
        v.convex = (n == 2) ? true : (leftOf(pp, v.p, pn) >= 0.0f);
        tab.push_back(v);
    }
    return base;
}

/* processObstacles (env.py:123, ALAN:209): upstream RVO2 builds a BSP tree over the obstacle edges
 * (KdTree::buildObstacleTreeRecursive, RVO2 Library v2.0.x).  The tree itself is not needed by a
 * brute-force edge scan, but building it SPLITS every edge that crosses the supporting line of the
 * edge chosen at a node, and the new vertices are appended to the simulator's vertex table, where
 * getObstacleVertex / getNextObstacleVertexNo (env.py:307-311) and the neighbour query see them.
 * This restates the published recursion and keeps only that side effect.  At a node the splitting edge
 * is the one minimising (max(left, right), min(left, right)) lexicographically, first minimum wins;
 * the left subtree is built before the right one, which fixes the numbering of the new vertices. */
void split_obstacles(std::vector<ObstVertex>& tab, const std::vector<int>& obs) {
    typedef std::pair<size_t, size_t> SS;
    const size_t n = obs.size();
    if (n == 0) return;
    size_t best = 0, minLeft = n, minRight = n;
    for (size_t i = 0; i < n; ++i) {
        size_t l = 0, r = 0;
        const V2 i1 = tab[obs[i]].p, i2 = tab[tab[obs[i]].next].p;
        for (size_t j = 0; j < n; ++j) {
            if (i == j) continue;
            const float a = leftOf(i1, i2, tab[obs[j]].p), b = leftOf(i1, i2, tab[tab[obs[j]].next].p);
            if (a >= -EPS && b >= -EPS) ++l;
            else if (a <= EPS && b <= EPS) ++r;
            else { ++l; ++r; }
            if (SS(std::max(l, r), std::min(l, r)) >= SS(std::max(minLeft, minRight), std::min(minLeft, minRight))) break;
        }
        if (SS(std::max(l, r), std::min(l, r)) < SS(std::max(minLeft, minRight), std::min(minLeft, minRight))) {
            minLeft = l; minRight = r; best = i;
        }
    }
    std::vector<int> left, right;
    const V2 i1 = tab[obs[best]].p, i2 = tab[tab[obs[best]].next].p;
    for (size_t j = 0; j < n; ++j) {
        if (j == best) continue;
        const int j1 = obs[j], j2 = tab[j1].next;
        const float a = leftOf(i1, i2, tab[j1].p), b = leftOf(i1, i2, tab[j2].p);
        if (a >= -EPS && b >= -EPS) left.push_back(j1);
        else if (a <= EPS && b <= EPS) right.push_back(j1);
        else { /* the edge j1 -> j2 crosses the line: cut it there */
            const float t = det(i2 - i1, tab[j1].p - i1) / det(i2 - i1, tab[j1].p - tab[j2].p);
            ObstVertex v;
            v.p = tab[j1].p + t * (tab[j2].p - tab[j1].p);
            v.prev = j1; v.next = j2; v.convex = true; v.unitDir = tab[j1].unitDir;
            const int id = (int)tab.size();
            tab.push_back(v);
            tab[j1].next = id; tab[j2].prev = id;
            if (a > 0.0f) { left.push_back(j1); right.push_back(id); }
            else { right.push_back(j1); left.push_back(id); }
        }
    }
    split_obstacles(tab, left);
    split_obstacles(tab, right);
}
void process_obstacles(std::vector<ObstVertex>& tab) {
    std::vector<int> all(tab.size());
    for (size_t i = 0; i < all.size(); ++i) all[i] = (int)i;
    split_obstacles(tab, all);
}

/* ------------------------------------------------------------------------------------------ */
/* one arena                                                                                   */
/* ------------------------------------------------------------------------------------------ */
typedef std::vector<std::pair<float, int> > NbList;

struct Arena {
    std::vector<V2> pos, vel, pref, newVel;
    std::vector<AgentParams> prm;
    std::vector<NbList> agentNb, obstNb;
    int n() const { return (int)pos.size(); }
    void resize(int n) {
        pos.resize(n); vel.resize(n); pref.resize(n); newVel.resize(n);
        prm.resize(n); agentNb.resize(n); obstNb.resize(n);
    }
};

/* App. A.2: K nearest agents with d^2 < neighborDist^2, ascending; candidates visited in index
 * order so ties resolve to the lower index; obstacle edges the agent is strictly to the right
 * of, with line and segment distance^2 < (tauObst*maxSpeed + r)^2, ascending */
void compute_neighbors(Arena& a, const std::vector<ObstVertex>& obst, int i, int obstCap,
                       uint64_t* overflow) {
    const AgentParams& P = a.prm[i];
    const V2 p = a.pos[i];
    NbList& on = a.obstNb[i];
    on.clear();
    {
        const float rangeSq = sqr(P.timeHorizonObst * P.maxSpeed + P.radius);
        for (int e = 0; e < (int)obst.size(); ++e) {
            const ObstVertex& o1 = obst[e];
            const ObstVertex& o2 = obst[o1.next];
            const float agentLeftOfLine = leftOf(o1.p, o2.p, p);
            const float distSqLine = sqr(agentLeftOfLine) / absSq(o2.p - o1.p);
            if (distSqLine < rangeSq && agentLeftOfLine < 0.0f) {
                const float distSq = distSqPointSegment(o1.p, o2.p, p);
                if (distSq < rangeSq) {
                    on.push_back(std::make_pair(distSq, e));
                    size_t k = on.size() - 1;
                    while (k != 0 && distSq < on[k - 1].first) {
                        on[k] = on[k - 1];
                        --k;
                    }
                    on[k] = std::make_pair(distSq, e);
                }
            }
        }
        if ((int)on.size() > obstCap) {
            if (overflow) *overflow += 1;
            on.resize(obstCap);
        }
    }
    NbList& an = a.agentNb[i];
    an.clear();
    if (P.maxNeighbors > 0) {
        float rangeSq = sqr(P.neighborDist);
        for (int j = 0; j < a.n(); ++j) {
            if (j == i) continue;
            const float distSq = absSq(p - a.pos[j]);
            if (distSq < rangeSq) {
                if ((int)an.size() < P.maxNeighbors) an.push_back(std::make_pair(distSq, j));
                size_t k = an.size() - 1;
                while (k != 0 && distSq < an[k - 1].first) {
                    an[k] = an[k - 1];
                    --k;
                }
                an[k] = std::make_pair(distSq, j);
                if ((int)an.size() == P.maxNeighbors) rangeSq = an.back().first;
            }
        }
    }
}

/* ---- diagnostics for tests/test_oracle_orca_definition.py: which branch of App. A.3 / A.4 / A.5 an agent-step took, and the
 * ORCA lines it built.  Thread-local (a shared counter written for every agent-step would serialise the multi-core
 * baseline on one cache line); a branch costs one increment.  Read through orc_debug_branches / orc_debug_capture_*. */
#define ORC_BRANCHES(X)                                                                                                   \
    X(OBST_EDGE) X(OBST_COVERED) X(OBST_COLL_LEFT_VERTEX) X(OBST_COLL_LEFT_VERTEX_NONCONVEX) X(OBST_COLL_RIGHT_VERTEX)    \
    X(OBST_COLL_RIGHT_VERTEX_SKIPPED) X(OBST_COLL_SEGMENT) X(OBST_OBLIQUE_LEFT) X(OBST_OBLIQUE_LEFT_NONCONVEX)            \
    X(OBST_OBLIQUE_RIGHT) X(OBST_OBLIQUE_RIGHT_NONCONVEX) X(OBST_LEGS_USUAL) X(OBST_LEFT_LEG_NONCONVEX)                   \
    X(OBST_RIGHT_LEG_NONCONVEX) X(OBST_LEFT_LEG_FOREIGN) X(OBST_RIGHT_LEG_FOREIGN) X(OBST_PROJ_LEFT_CIRCLE)               \
    X(OBST_PROJ_RIGHT_CIRCLE) X(OBST_PROJ_CUTOFF) X(OBST_PROJ_LEFT_LEG) X(OBST_PROJ_LEFT_LEG_FOREIGN_SKIPPED)             \
    X(OBST_PROJ_RIGHT_LEG) X(OBST_PROJ_RIGHT_LEG_FOREIGN_SKIPPED) X(AGENT_CUTOFF_CIRCLE) X(AGENT_LEG_LEFT)                \
    X(AGENT_LEG_RIGHT) X(AGENT_COLLISION) X(LP2_OPT_INSIDE_DISC) X(LP2_OPT_CLAMPED) X(LP2_DIR_OPT) X(LP2_LINE_SATISFIED)  \
    X(LP2_LINE_VIOLATED) X(LP2_FAILED) X(LP1_DISC_NEGATIVE) X(LP1_PARALLEL_FAIL) X(LP1_PARALLEL_OK) X(LP1_BOUND_RIGHT)    \
    X(LP1_BOUND_LEFT) X(LP1_EMPTIED) X(LP1_DIR_OPT_RIGHT) X(LP1_DIR_OPT_LEFT) X(LP1_CLAMP_LEFT) X(LP1_CLAMP_RIGHT)        \
    X(LP1_INTERIOR) X(LP3_ENTERED) X(LP3_ENTERED_WITH_OBST_LINES) X(LP3_LINE_WITHIN_DIST) X(LP3_LINE_BEYOND_DIST)         \
    X(LP3_PARALLEL_SAME_DIR) X(LP3_PARALLEL_OPPOSITE) X(LP3_PROJECTED) X(LP3_LP2_FAILED_RESTORED) X(LP3_LP2_OK)
enum {
#define X(n) BR_##n,
    ORC_BRANCHES(X)
#undef X
    BR__COUNT
};
static const char* const g_br_names[BR__COUNT] = {
#define X(n) #n,
    ORC_BRANCHES(X)
#undef X
};
static thread_local uint64_t g_br[BR__COUNT];
#define BR(n) (++g_br[BR_##n])
struct Capture {                     /* the lines of ONE agent-step (orc_debug_capture selects it) */
    int arena = -1, agent = -1, armed = 0;
    int numObst = 0, nObstNb = 0, fail = 0, nl = 0;
    std::vector<Line> lines;         /* obstacle lines first */
    std::vector<int> edge;           /* per obstacle LINE: the edge id (processed table) it came from */
    std::vector<int> tag;            /* per obstacle NEIGHBOUR, in list order: edge id << 8 | the BR_OBST_* outcome */
};
static thread_local Capture g_cap;
static thread_local int g_cur_arena = -1;   /* set by the stepping loops while a capture is armed */

/* App. A.5 LP1 */
bool lp1(const std::vector<Line>& lines, int lineNo, float radius, V2 opt, bool dirOpt, V2& result) {
    const Line& L = lines[lineNo];
    const float dp = dot(L.point, L.dir);
    const float disc = sqr(dp) + sqr(radius) - absSq(L.point);
    if (disc < 0.0f) { BR(LP1_DISC_NEGATIVE); return false; }
    const float sq = std::sqrt(disc);
    float tLeft = -dp - sq;
    float tRight = -dp + sq;
    for (int j = 0; j < lineNo; ++j) {
        const float den = det(L.dir, lines[j].dir);
        const float num = det(lines[j].dir, L.point - lines[j].point);
        if (std::fabs(den) <= EPS) {
            if (num < 0.0f) { BR(LP1_PARALLEL_FAIL); return false; }
            BR(LP1_PARALLEL_OK);
            continue;
        }
        const float t = num / den;
        if (den >= 0.0f) { BR(LP1_BOUND_RIGHT); tRight = std::min(tRight, t); }
        else { BR(LP1_BOUND_LEFT); tLeft = std::max(tLeft, t); }
        if (tLeft > tRight) { BR(LP1_EMPTIED); return false; }
    }
    if (dirOpt) {
        if (dot(opt, L.dir) > 0.0f) { BR(LP1_DIR_OPT_RIGHT); result = L.point + tRight * L.dir; }
        else { BR(LP1_DIR_OPT_LEFT); result = L.point + tLeft * L.dir; }
    } else {
        const float t = dot(L.dir, opt - L.point);
        if (t < tLeft) { BR(LP1_CLAMP_LEFT); result = L.point + tLeft * L.dir; }
        else if (t > tRight) { BR(LP1_CLAMP_RIGHT); result = L.point + tRight * L.dir; }
        else { BR(LP1_INTERIOR); result = L.point + t * L.dir; }
    }
    return true;
}

/* App. A.5 LP2 */
int lp2(const std::vector<Line>& lines, float radius, V2 opt, bool dirOpt, V2& result) {
    if (dirOpt) { BR(LP2_DIR_OPT); result = opt * radius; }
    else if (absSq(opt) > sqr(radius)) { BR(LP2_OPT_CLAMPED); result = normalize(opt) * radius; }
    else { BR(LP2_OPT_INSIDE_DISC); result = opt; }
    for (int i = 0; i < (int)lines.size(); ++i) {
        if (det(lines[i].dir, lines[i].point - result) > 0.0f) {
            BR(LP2_LINE_VIOLATED);
            const V2 tmp = result;
            if (!lp1(lines, i, radius, opt, dirOpt, result)) {
                BR(LP2_FAILED);
                result = tmp;
                return i;
            }
        } else {
            BR(LP2_LINE_SATISFIED);
        }
    }
    return (int)lines.size();
}

/* App. A.5 LP3 */
void lp3(const std::vector<Line>& lines, int numObst, int begin, float radius, V2& result) {
    float distance = 0.0f;
    BR(LP3_ENTERED);
    if (numObst > 0) BR(LP3_ENTERED_WITH_OBST_LINES);
    for (int i = begin; i < (int)lines.size(); ++i) {
        if (det(lines[i].dir, lines[i].point - result) > distance) {
            BR(LP3_LINE_BEYOND_DIST);
            static thread_local std::vector<Line> proj;
            proj.assign(lines.begin(), lines.begin() + numObst);
            for (int j = numObst; j < i; ++j) {
                Line l;
                const float d = det(lines[i].dir, lines[j].dir);
                if (std::fabs(d) <= EPS) {
                    if (dot(lines[i].dir, lines[j].dir) > 0.0f) { BR(LP3_PARALLEL_SAME_DIR); continue; }
                    BR(LP3_PARALLEL_OPPOSITE);
                    l.point = 0.5f * (lines[i].point + lines[j].point);
                } else {
                    BR(LP3_PROJECTED);
                    l.point = lines[i].point +
                              (det(lines[j].dir, lines[i].point - lines[j].point) / d) * lines[i].dir;
                }
                l.dir = normalize(lines[j].dir - lines[i].dir);
                proj.push_back(l);
            }
            const V2 tmp = result;
            if (lp2(proj, radius, mk(-lines[i].dir.y, lines[i].dir.x), true, result) < (int)proj.size()) {
                BR(LP3_LP2_FAILED_RESTORED);
                result = tmp;
            } else {
                BR(LP3_LP2_OK);
            }
            distance = det(lines[i].dir, lines[i].point - result);
        } else {
            BR(LP3_LINE_WITHIN_DIST);
        }
    }
}

/* diagnostics: how often the infeasible path (LP3) is taken, and how many lines it re-solves */
static thread_local uint64_t g_dbg[4] = {0, 0, 0, 0};  /* per thread: a shared counter written for every agent-step
                                                         * serialises the multi-core baseline on one cache line */

/* App. A.3 + A.4 + the LP call sequence of A.5 */
void compute_new_velocity(Arena& a, const std::vector<ObstVertex>& obst, int i, float timeStep) {
    const AgentParams& P = a.prm[i];
    const V2 pos = a.pos[i], vel = a.vel[i];
    static thread_local std::vector<Line> lines;  /* reused: no allocation per agent-step */
    lines.clear();
    const float invTO = 1.0f / P.timeHorizonObst;
    const float R = P.radius;

    const bool cap = g_cap.armed && g_cap.agent == i && g_cap.arena == g_cur_arena;
    if (cap) { g_cap.edge.clear(); g_cap.tag.clear(); }
    for (size_t n = 0; n < a.obstNb[i].size(); ++n) {
        int i1 = a.obstNb[i][n].second;
        const int edge_id = i1;
        int i2 = obst[i1].next;
        /* the outcome of this neighbour: counted, and recorded when this agent-step is being captured */
        auto outcome = [&](int br, bool emitted) {
            ++g_br[br];
            if (cap) { g_cap.tag.push_back((edge_id << 8) | br); if (emitted) g_cap.edge.push_back(edge_id); }
        };
        BR(OBST_EDGE);
        const V2 rp1 = obst[i1].p - pos;
        const V2 rp2 = obst[i2].p - pos;
        bool covered = false;
        for (size_t j = 0; j < lines.size(); ++j) {
            if (det(invTO * rp1 - lines[j].point, lines[j].dir) - invTO * R >= -EPS &&
                det(invTO * rp2 - lines[j].point, lines[j].dir) - invTO * R >= -EPS) {
                covered = true;
                break;
            }
        }
        if (covered) { outcome(BR_OBST_COVERED, false); continue; }
        const float distSq1 = absSq(rp1), distSq2 = absSq(rp2), radiusSq = sqr(R);
        const V2 ov = obst[i2].p - obst[i1].p;
        const float s = dot(-rp1, ov) / absSq(ov);
        const float distSqLine = absSq(-rp1 - s * ov);
        Line line;
        if (s < 0.0f && distSq1 <= radiusSq) {
            if (obst[i1].convex) {
                line.point = mk(0.0f, 0.0f);
                line.dir = normalize(mk(-rp1.y, rp1.x));
                lines.push_back(line);
                outcome(BR_OBST_COLL_LEFT_VERTEX, true);
            } else {
                outcome(BR_OBST_COLL_LEFT_VERTEX_NONCONVEX, false);
            }
            continue;
        } else if (s > 1.0f && distSq2 <= radiusSq) {
            if (obst[i2].convex && det(rp2, obst[i2].unitDir) >= 0.0f) {
                line.point = mk(0.0f, 0.0f);
                line.dir = normalize(mk(-rp2.y, rp2.x));
                lines.push_back(line);
                outcome(BR_OBST_COLL_RIGHT_VERTEX, true);
            } else {
                outcome(BR_OBST_COLL_RIGHT_VERTEX_SKIPPED, false);
            }
            continue;
        } else if (s >= 0.0f && s < 1.0f && distSqLine <= radiusSq) {
            line.point = mk(0.0f, 0.0f);
            line.dir = -obst[i1].unitDir;
            lines.push_back(line);
            outcome(BR_OBST_COLL_SEGMENT, true);
            continue;
        }
        V2 leftLeg, rightLeg;
        if (s < 0.0f && distSqLine <= radiusSq) {
            if (!obst[i1].convex) { outcome(BR_OBST_OBLIQUE_LEFT_NONCONVEX, false); continue; }
            BR(OBST_OBLIQUE_LEFT);
            i2 = i1;
            const float leg1 = std::sqrt(distSq1 - radiusSq);
            leftLeg = vdiv(mk(rp1.x * leg1 - rp1.y * R, rp1.x * R + rp1.y * leg1), distSq1);
            rightLeg = vdiv(mk(rp1.x * leg1 + rp1.y * R, -rp1.x * R + rp1.y * leg1), distSq1);
        } else if (s > 1.0f && distSqLine <= radiusSq) {
            if (!obst[i2].convex) { outcome(BR_OBST_OBLIQUE_RIGHT_NONCONVEX, false); continue; }
            BR(OBST_OBLIQUE_RIGHT);
            i1 = i2;
            const float leg2 = std::sqrt(distSq2 - radiusSq);
            leftLeg = vdiv(mk(rp2.x * leg2 - rp2.y * R, rp2.x * R + rp2.y * leg2), distSq2);
            rightLeg = vdiv(mk(rp2.x * leg2 + rp2.y * R, -rp2.x * R + rp2.y * leg2), distSq2);
        } else {
            BR(OBST_LEGS_USUAL);
            if (obst[i1].convex) {
                const float leg1 = std::sqrt(distSq1 - radiusSq);
                leftLeg = vdiv(mk(rp1.x * leg1 - rp1.y * R, rp1.x * R + rp1.y * leg1), distSq1);
            } else {
                BR(OBST_LEFT_LEG_NONCONVEX);
                leftLeg = -obst[i1].unitDir;
            }
            if (obst[i2].convex) {
                const float leg2 = std::sqrt(distSq2 - radiusSq);
                rightLeg = vdiv(mk(rp2.x * leg2 + rp2.y * R, -rp2.x * R + rp2.y * leg2), distSq2);
            } else {
                BR(OBST_RIGHT_LEG_NONCONVEX);
                rightLeg = obst[i1].unitDir;
            }
        }
        const int leftNb = obst[i1].prev;
        bool leftForeign = false, rightForeign = false;
        if (obst[i1].convex && det(leftLeg, -obst[leftNb].unitDir) >= 0.0f) {
            BR(OBST_LEFT_LEG_FOREIGN);
            leftLeg = -obst[leftNb].unitDir;
            leftForeign = true;
        }
        if (obst[i2].convex && det(rightLeg, obst[i2].unitDir) <= 0.0f) {
            BR(OBST_RIGHT_LEG_FOREIGN);
            rightLeg = obst[i2].unitDir;
            rightForeign = true;
        }
        const V2 leftCut = invTO * (obst[i1].p - pos);
        const V2 rightCut = invTO * (obst[i2].p - pos);
        const V2 cutVec = rightCut - leftCut;
        const bool same = (i1 == i2);
        const float t = same ? 0.5f : dot(vel - leftCut, cutVec) / absSq(cutVec);
        const float tLeft = dot(vel - leftCut, leftLeg);
        const float tRight = dot(vel - rightCut, rightLeg);
        if ((t < 0.0f && tLeft < 0.0f) || (same && tLeft < 0.0f && tRight < 0.0f)) {
            const V2 unitW = normalize(vel - leftCut);
            line.dir = mk(unitW.y, -unitW.x);
            line.point = leftCut + R * invTO * unitW;
            lines.push_back(line);
            outcome(BR_OBST_PROJ_LEFT_CIRCLE, true);
            continue;
        } else if (t > 1.0f && tRight < 0.0f) {
            const V2 unitW = normalize(vel - rightCut);
            line.dir = mk(unitW.y, -unitW.x);
            line.point = rightCut + R * invTO * unitW;
            lines.push_back(line);
            outcome(BR_OBST_PROJ_RIGHT_CIRCLE, true);
            continue;
        }
        const float dCut = (t < 0.0f || t > 1.0f || same) ? FINF : absSq(vel - (leftCut + t * cutVec));
        const float dLeft = (tLeft < 0.0f) ? FINF : absSq(vel - (leftCut + tLeft * leftLeg));
        const float dRight = (tRight < 0.0f) ? FINF : absSq(vel - (rightCut + tRight * rightLeg));
        if (dCut <= dLeft && dCut <= dRight) {
            line.dir = -obst[i1].unitDir;
            line.point = leftCut + R * invTO * mk(-line.dir.y, line.dir.x);
            lines.push_back(line);
            outcome(BR_OBST_PROJ_CUTOFF, true);
            continue;
        } else if (dLeft <= dRight) {
            if (leftForeign) { outcome(BR_OBST_PROJ_LEFT_LEG_FOREIGN_SKIPPED, false); continue; }
            line.dir = leftLeg;
            line.point = leftCut + R * invTO * mk(-line.dir.y, line.dir.x);
            lines.push_back(line);
            outcome(BR_OBST_PROJ_LEFT_LEG, true);
            continue;
        } else {
            if (rightForeign) { outcome(BR_OBST_PROJ_RIGHT_LEG_FOREIGN_SKIPPED, false); continue; }
            line.dir = -rightLeg;
            line.point = rightCut + R * invTO * mk(-line.dir.y, line.dir.x);
            lines.push_back(line);
            outcome(BR_OBST_PROJ_RIGHT_LEG, true);
        }
    }
    const int numObstLines = (int)lines.size();

    const float invT = 1.0f / P.timeHorizon;
    for (size_t n = 0; n < a.agentNb[i].size(); ++n) {
        const int o = a.agentNb[i][n].second;
        const V2 rp = a.pos[o] - pos;
        const V2 rv = vel - a.vel[o];
        const float distSq = absSq(rp);
        const float cr = R + a.prm[o].radius;
        const float crSq = sqr(cr);
        Line line;
        V2 u;
        if (distSq > crSq) {
            const V2 w = rv - invT * rp;
            const float wLenSq = absSq(w);
            const float dp1 = dot(w, rp);
            if (dp1 < 0.0f && sqr(dp1) > crSq * wLenSq) {
                BR(AGENT_CUTOFF_CIRCLE);
                const float wLen = std::sqrt(wLenSq);
                const V2 unitW = vdiv(w, wLen);
                line.dir = mk(unitW.y, -unitW.x);
                u = (cr * invT - wLen) * unitW;
            } else {
                const float leg = std::sqrt(distSq - crSq);
                if (det(rp, w) > 0.0f) {
                    BR(AGENT_LEG_LEFT);
                    line.dir = vdiv(mk(rp.x * leg - rp.y * cr, rp.x * cr + rp.y * leg), distSq);
                } else {
                    BR(AGENT_LEG_RIGHT);
                    line.dir = -vdiv(mk(rp.x * leg + rp.y * cr, -rp.x * cr + rp.y * leg), distSq);
                }
                const float dp2 = dot(rv, line.dir);
                u = dp2 * line.dir - rv;
            }
        } else {
            BR(AGENT_COLLISION);
            const float invDt = 1.0f / timeStep;
            const V2 w = rv - invDt * rp;
            const float wLen = vabs(w);
            const V2 unitW = vdiv(w, wLen);
            line.dir = mk(unitW.y, -unitW.x);
            u = (cr * invDt - wLen) * unitW;
        }
        line.point = vel + 0.5f * u;
        lines.push_back(line);
    }
    V2 nv = mk(0.0f, 0.0f);
    const int fail = lp2(lines, P.maxSpeed, a.pref[i], false, nv);
    if (cap) {
        g_cap.lines = lines; g_cap.numObst = numObstLines; g_cap.nObstNb = (int)a.obstNb[i].size(); g_cap.fail = fail;
        g_cap.nl = (int)lines.size(); g_cap.armed = 2;
    }
    g_dbg[0] += 1;
    if (fail < (int)lines.size()) { g_dbg[1] += 1; g_dbg[2] += lines.size() - fail; lp3(lines, numObstLines, fail, P.maxSpeed, nv); }
    a.newVel[i] = nv;
}

/* App. A.1 doStep: two-phase */
void do_step(Arena& a, const std::vector<ObstVertex>& obst, float timeStep, int obstCap,
             uint64_t* overflow) {
    for (int i = 0; i < a.n(); ++i) {
        compute_neighbors(a, obst, i, obstCap, overflow);
        compute_new_velocity(a, obst, i, timeStep);
    }
    for (int i = 0; i < a.n(); ++i) {
        a.vel[i] = a.newVel[i];
        a.pos[i] = a.pos[i] + a.vel[i] * timeStep;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* deterministic fp64 helpers shared (by specification, not by code) with the HIP side         */
/* ------------------------------------------------------------------------------------------ */
/* sin/cos by quadrant reduction + degree-13/14 polynomials; every operation is a single IEEE
 * fp64 add/mul/floor so CPU and GPU agree bit for bit.  |a| up to ~1e5; error <~ 2e-16. */
void sincos64(double a, double* s, double* c) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00; /* 33 leading bits of pi/2 */
    const double PIO2_LO = 6.07710050650619224932e-11; /* pi/2 - PIO2_HI          */
    const double kd = std::floor(a * TWO_OVER_PI + 0.5);
    const double r = (a - kd * PIO2_HI) - kd * PIO2_LO;
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double sp = r + (z * r) * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))));
    const double cp = 1.0 - (0.5 * z - (z * z) * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6))))));
    const long long q = (long long)kd;
    switch ((int)(q & 3)) {
        case 0: *s = sp; *c = cp; break;
        case 1: *s = cp; *c = -sp; break;
        case 2: *s = -sp; *c = -cp; break;
        default: *s = -cp; *c = sp; break;
    }
}

/* e^x by k = round(x/ln2), r = x - k ln2 (two-part), a degree-13 Taylor polynomial in Horner form
 * and an exact scaling by 2^k: single IEEE operations only, |x| <~ 700, error <~ 2e-16 relative */
double exp64(double x) {
    const double kd = std::floor(x * 1.44269504088896338700e+00 + 0.5);
    const double r = (x - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10;
    double pl = 1.0 / 6227020800.0;
    pl = pl * r + 1.0 / 479001600.0;
    pl = pl * r + 1.0 / 39916800.0;
    pl = pl * r + 1.0 / 3628800.0;
    pl = pl * r + 1.0 / 362880.0;
    pl = pl * r + 1.0 / 40320.0;
    pl = pl * r + 1.0 / 5040.0;
    pl = pl * r + 1.0 / 720.0;
    pl = pl * r + 1.0 / 120.0;
    pl = pl * r + 1.0 / 24.0;
    pl = pl * r + 1.0 / 6.0;
    pl = pl * r + 0.5;
    pl = pl * r + 1.0;
    pl = pl * r + 1.0;
    const long long k = (long long)kd;
    uint64_t bits = (uint64_t)(k + 1023) << 52;
    double scale;
    std::memcpy(&scale, &bits, 8);
    return pl * scale;
}

/* numpy's float64 sum for short arrays (pairwise_sum: < 8 sequential; otherwise 8 accumulators
 * combined as a tree, then the tail) -- what `np.sum(ps)` at ALAN_true.py:582 computes */
double np_sum(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

/* env.py:156-162 comp_pref_vel: (cos, sin) of atan2(goal - pos) in fp64 == the normalised
 * difference; atan2(0,0) = 0 -> (1,0).  Positions are fp32 (the simulator's), targets fp64 (the
 * reference keeps them as Python floats, env.py:94, ALAN_true.py:186-187). */
void pref_dir64(float px, float py, double gx, double gy, double* ox, double* oy) {
    const double dx = gx - (double)px;
    const double dy = gy - (double)py;
    if (dx == 0.0 && dy == 0.0) {
        *ox = 1.0;
        *oy = 0.0;
        return;
    }
    const double len = std::sqrt(dx * dx + dy * dy);
    *ox = dx / len;
    *oy = dy / len;
}

/* Philox4x32-10 (Salmon et al., SC'11) */
void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                uint32_t* out) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* purposes of a draw */
enum { RNG_POS = 0, RNG_HEADING = 1, RNG_GOAL = 2, RNG_REGOAL = 3, RNG_RESET = 4, RNG_ALAN = 5 };

/* two uniform doubles in [0,1) with 53 random bits each, keyed by (seed, arena, agent, purpose, seq) */
void rng2(uint64_t seed, int64_t arena, int agent, int purpose, uint32_t seq, double* u0, double* u1) {
    uint32_t w[4];
    const uint64_t g = (uint64_t)arena;
    philox4x32((uint32_t)g, (uint32_t)agent, (uint32_t)purpose, seq, (uint32_t)seed,
               (uint32_t)(seed >> 32) + (uint32_t)(g >> 32), w);
    const double k = 1.0 / 9007199254740992.0; /* 2^-53 */
    *u0 = (double)(((uint64_t)(w[0] >> 5) << 26) | (uint64_t)(w[1] >> 6)) * k;
    *u1 = (double)(((uint64_t)(w[2] >> 5) << 26) | (uint64_t)(w[3] >> 6)) * k;
}
inline double uniform64(double a, double b, double u) { return a + (b - a) * u; } /* random.uniform */

/* env.py:321-332: ray i ends at (nd cos th, -nd sin th), th = i * 2pi/16 */
void ray_table(double nd, double* out) {
    const double d = 2.0 * M_PI / ORC_N_RAYS;
    for (int i = 0; i < ORC_N_RAYS; ++i) {
        const double th = i * d;
        out[2 * i] = nd * std::cos(th);
        out[2 * i + 1] = -nd * std::sin(th);
    }
}
/* env.py:335-350: 8 chords of the circle of radius r, clockwise in math coordinates, closed */
void octagon_table(double r, double* out) {
    const double d = 2.0 * M_PI / ORC_N_OCT;
    double first[2] = {r * std::cos(0.0), -r * std::sin(0.0)};
    double cur[2] = {first[0], first[1]};
    int k = 0;
    for (int i = 1; i < ORC_N_OCT; ++i) {
        const double th = i * d;
        const double nx = r * std::cos(th), ny = -r * std::sin(th);
        out[4 * k] = cur[0]; out[4 * k + 1] = cur[1]; out[4 * k + 2] = nx; out[4 * k + 3] = ny;
        ++k;
        cur[0] = nx; cur[1] = ny;
    }
    out[4 * k] = cur[0]; out[4 * k + 1] = cur[1]; out[4 * k + 2] = first[0]; out[4 * k + 3] = first[1];
}

/* ------------------------------------------------------------------------------------------ */
/* laser observation (utils.py), templated on the arithmetic type                              */
/* ------------------------------------------------------------------------------------------ */
/* utils.py:5-40 with p0 = (0,0), p1 = ray end */
template <class T>
inline bool ray_hit(T rx, T ry, T p2x, T p2y, T p3x, T p3y, T* d, T* hx, T* hy) {
    const T s10x = rx - T(0), s10y = ry - T(0);
    const T s32x = p3x - p2x, s32y = p3y - p2y;
    const T denom = s10x * s32y - s32x * s10y;
    if (denom == T(0)) return false;
    const bool dpos = denom > T(0);
    const T s02x = T(0) - p2x, s02y = T(0) - p2y;
    const T s_numer = s10x * s02y - s10y * s02x;
    if ((s_numer < T(0)) == dpos) return false;
    const T t_numer = s32x * s02y - s32y * s02x;
    if ((t_numer < T(0)) == dpos) return false;
    if (((s_numer > denom) == dpos) || ((t_numer > denom) == dpos)) return false;
    const T t = t_numer / denom;
    const T px = T(0) + t * s10x;
    const T py = T(0) + t * s10y;
    *d = std::sqrt(px * px + py * py);
    *hx = px;
    *hy = py;
    return true;
}

template <class T>
struct Seg {
    T x1, y1, x2, y2, vx, vy;
};

/* utils.py:42-113.  (c, s) = (cos th, sin th), th = -atan2(orientation).  Ties between equal
 * distances resolve to the first segment (the reference's argsort is unspecified on ties). */
/* How close a ray's answer is to flipping under rounding (test diagnostic, not part of the reference): for every
 * segment the distance of its (s, t) = (s_numer, t_numer) / denom from the boundary of the accepted square [0,1]^2
 * (utils.py:21-31), and the gap between the two nearest accepted hits relative to the ray length; the minimum. */
template <class T>
inline T ray_flip_margin(T rx, T ry, const Seg<T>* rot, int m) {
    T margin = std::numeric_limits<T>::infinity(), d1 = margin, d2 = margin;
    const T len = std::sqrt(rx * rx + ry * ry);
    for (int k = 0; k < m; ++k) {
        const T s32x = rot[k].x2 - rot[k].x1, s32y = rot[k].y2 - rot[k].y1;
        const T denom = rx * s32y - s32x * ry;
        if (denom == T(0)) continue;
        const T s02x = T(0) - rot[k].x1, s02y = T(0) - rot[k].y1;
        const T sp = (rx * s02y - ry * s02x) / denom, tp = (s32x * s02y - s32y * s02x) / denom;
        const T es = std::max(std::max(-sp, sp - T(1)), T(0)), et = std::max(std::max(-tp, tp - T(1)), T(0));
        T b;
        if (es > T(0) || et > T(0)) b = std::max(es, et);                       /* outside: distance to the square */
        else b = std::min(std::min(sp, T(1) - sp), std::min(tp, T(1) - tp));    /* inside: distance to its boundary */
        margin = std::min(margin, b);
        if (es == T(0) && et == T(0)) {
            const T d = tp * len;
            if (d < d1) { d2 = d1; d1 = d; } else if (d < d2) d2 = d;
        }
    }
    if (d2 < std::numeric_limits<T>::infinity()) margin = std::min(margin, (d2 - d1) / len);
    return margin;
}

template <class T>
void comp_laser(const T* rays, const Seg<T>* segs, int m, T c, T s, T* out, double* margin = nullptr) {
    static thread_local std::vector<Seg<T> > rot;  /* reused: no allocation per agent-step (the multi-core baseline) */
    rot.resize(m);
    for (int k = 0; k < m; ++k) {
        const Seg<T>& g = segs[k];
        const T lvx = g.x1 + g.vx, lvy = g.y1 + g.vy;           /* utils.py:57 */
        Seg<T> r;
        r.x1 = c * g.x1 - s * g.y1; r.y1 = s * g.x1 + c * g.y1;  /* utils.py:59 */
        r.x2 = c * g.x2 - s * g.y2; r.y2 = s * g.x2 + c * g.y2;  /* utils.py:60 */
        const T rvx = c * lvx - s * lvy, rvy = s * lvx + c * lvy; /* utils.py:61 */
        r.vx = rvx - r.x1; r.vy = rvy - r.y1;                    /* utils.py:62 */
        rot[k] = r;
    }
    for (int i = 0; i < ORC_N_RAYS; ++i) {
        T best = std::numeric_limits<T>::infinity();
        T bx = T(0), by = T(0);
        int bi = -1;
        for (int k = 0; k < m; ++k) {
            T d, hx, hy;
            if (ray_hit<T>(rays[2 * i], rays[2 * i + 1], rot[k].x1, rot[k].y1, rot[k].x2, rot[k].y2,
                           &d, &hx, &hy)) {
                if (d < best) { best = d; bx = hx; by = hy; bi = k; }
            }
        }
        T vx = T(0), vy = T(0);
        if (bi >= 0 && !(bx == T(0) && by == T(0))) { vx = rot[bi].vx; vy = rot[bi].vy; } /* utils.py:103 */
        out[4 * i] = bx; out[4 * i + 1] = by; out[4 * i + 2] = vx; out[4 * i + 3] = vy;
        if (margin) margin[i] = (double)ray_flip_margin<T>(rays[2 * i], rays[2 * i + 1], rot.data(), m);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* batch environment                                                                           */
/* ------------------------------------------------------------------------------------------ */
struct Env {
    orc_config cfg;
    std::vector<ObstVertex> obst;                   /* the world every arena shares ...                    */
    std::vector<std::vector<ObstVertex> > obst_a;   /* ... or one world per arena (ALAN:359-372: every simulator
                                                       draws its own blocks); ids are local to the arena's table */
    const std::vector<ObstVertex>& tab(int a) const { return obst_a.empty() ? obst : obst_a[a]; }
    std::vector<Arena> arenas;
    std::vector<double> goal_x, goal_y, goal2_x, goal2_y;  /* fp64 like the reference's target tuples */
    std::vector<float> reward, obs;
    std::vector<double> reward64, obs64;
    std::vector<double> obs_margin;   /* [A*N*16], ORC_PREC_F64 only: ray_flip_margin of every ray (test diagnostic) */
    std::vector<int32_t> agent_done, arrive_step, regoal_count;
    std::vector<int32_t> step_count, arena_done, episode;
    std::vector<double> rl64;   /* scratch: action-rotated preferred direction, [A*N*2] */
    std::vector<double> pf64;   /* scratch: goal direction before the step, [A*N*2]     */
    std::vector<orc_stats> st;  /* per arena, summed on read                             */
    std::vector<uint64_t> frozen_steps, last_episode; /* [A]: steps sat out under FREEZE; (length << 32 | agents arrived) of the last finished episode */
    /* ALAN online learning state (ALAN_true.py:30-49, 141-142) */
    int n_actions = 0;
    double alan_temp = 0.2, alan_window = 2.0, alan_dt = 1.0 / 60.0;  /* Python floats of the reference */
    std::vector<double> act_c, act_s;          /* unit action vectors                    */
    std::vector<double> alan_w, alan_t;        /* [A*N*nA]                               */
    std::vector<int32_t> alan_action;          /* [A*N]                                  */
    double rays64[32], oct64[32];
    float rays32[32], oct32[32];
    int A() const { return cfg.n_arenas; }
    int N() const { return cfg.n_agents; }
};

void env_tables(Env& e) {
    ray_table((double)e.cfg.neighbor_dist, e.rays64);
    octagon_table((double)e.cfg.radius, e.oct64);
    for (int i = 0; i < 32; ++i) { e.rays32[i] = (float)e.rays64[i]; e.oct32[i] = (float)e.oct64[i]; }
}

void set_pref_toward_goal(Env& e, int a) {
    Arena& ar = e.arenas[a];
    const int N = e.N();
    for (int i = 0; i < N; ++i) { /* env.py:151-154 */
        double dx, dy;
        pref_dir64(ar.pos[i].x, ar.pos[i].y, e.goal_x[a * N + i], e.goal_y[a * N + i], &dx, &dy);
        ar.pref[i] = mk((float)dx, (float)dy);
    }
}

/* env.py:231-318 (_get_obs, _obs_neighbor_agent_lines, _obs_obstacle_lines) for one agent */
template <class T>
void agent_obs(Env& e, int a, int i, T* out, double* margin = nullptr) {
    Arena& ar = e.arenas[a];
    const int N = e.N();
    const T* oct = (sizeof(T) == 8) ? (const T*)(const void*)e.oct64 : (const T*)(const void*)e.oct32;
    const T* rays = (sizeof(T) == 8) ? (const T*)(const void*)e.rays64 : (const T*)(const void*)e.rays32;
    static thread_local std::vector<Seg<T> > segs;
    segs.clear();
    const V2 me = ar.pos[i];
    for (size_t k = 0; k < ar.agentNb[i].size(); ++k) { /* env.py:283-294 */
        const int nb = ar.agentNb[i][k].second;
        const T rx = (T)ar.pos[nb].x - (T)me.x, ry = (T)ar.pos[nb].y - (T)me.y;
        for (int s = 0; s < ORC_N_OCT; ++s) {
            Seg<T> g;
            g.x1 = oct[4 * s] + rx; g.y1 = oct[4 * s + 1] + ry;
            g.x2 = oct[4 * s + 2] + rx; g.y2 = oct[4 * s + 3] + ry;
            g.vx = (T)ar.vel[nb].x; g.vy = (T)ar.vel[nb].y; /* env.py:252 */
            segs.push_back(g);
        }
    }
    for (size_t k = 0; k < ar.obstNb[i].size(); ++k) { /* env.py:305-315 */
        const std::vector<ObstVertex>& ob = e.tab(a);
        const int v1 = ar.obstNb[i][k].second, v2 = ob[v1].next;
        Seg<T> g;
        g.x1 = (T)ob[v1].p.x - (T)me.x; g.y1 = (T)ob[v1].p.y - (T)me.y;
        g.x2 = (T)ob[v2].p.x - (T)me.x; g.y2 = (T)ob[v2].p.y - (T)me.y;
        g.vx = T(0); g.vy = T(0);
        segs.push_back(g);
    }
    if (segs.empty()) { /* env.py:267 */
        for (int k = 0; k < ORC_OBS_DIM; ++k) out[k] = T(0);
        if (margin) for (int k = 0; k < ORC_N_RAYS; ++k) margin[k] = std::numeric_limits<double>::infinity();
        return;
    }
    double ox, oy; /* env.py:236: orientation = comp_pref_vel (current position, current target) */
    pref_dir64(me.x, me.y, e.goal_x[a * N + i], e.goal_y[a * N + i], &ox, &oy);
    /* utils.py:48-51: th = -atan2(oy, ox); cos th = ox, sin th = -oy for a unit vector */
    const T c = (T)ox, s = (T)(-oy);
    comp_laser<T>(rays, segs.data(), (int)segs.size(), c, s, out, margin);
}

void arena_obs(Env& e, int a, int prec) {
    const int N = e.N();
    for (int i = 0; i < N; ++i) {
        if (prec == ORC_PREC_F64) {
            agent_obs<double>(e, a, i, &e.obs64[((size_t)a * N + i) * ORC_OBS_DIM],
                              &e.obs_margin[((size_t)a * N + i) * ORC_N_RAYS]);
            for (int k = 0; k < ORC_OBS_DIM; ++k)
                e.obs[((size_t)a * N + i) * ORC_OBS_DIM + k] = (float)e.obs64[((size_t)a * N + i) * ORC_OBS_DIM + k];
        } else {
            agent_obs<float>(e, a, i, &e.obs[((size_t)a * N + i) * ORC_OBS_DIM]);
        }
    }
}

/* build-defined statistic: overlapping pairs / wall overlaps after the update */
void arena_collisions(Env& e, int a) {
    Arena& ar = e.arenas[a];
    const int N = e.N();
    uint64_t pairs = 0, walls = 0;
    for (int i = 0; i < N; ++i) {
        for (int j = i + 1; j < N; ++j) {
            const float cr = ar.prm[i].radius + ar.prm[j].radius;
            if (absSq(ar.pos[i] - ar.pos[j]) < sqr(cr)) ++pairs;
        }
        bool hit = false;
        const std::vector<ObstVertex>& ob = e.tab(a);
        for (size_t k = 0; k < ob.size(); ++k) {
            const ObstVertex& o1 = ob[k];
            if (distSqPointSegment(o1.p, ob[o1.next].p, ar.pos[i]) < sqr(ar.prm[i].radius)) hit = true;
        }
        if (hit) ++walls;
    }
    e.st[a].collisions += pairs;
    e.st[a].obst_collisions += walls;
}

/* env.py:352-365 / ALAN:547-566 / bench regoal; returns 1 when every agent is done */
int arena_done_test(Env& e, int a) {
    Arena& ar = e.arenas[a];
    const int N = e.N();
    const orc_config& c = e.cfg;
    int all = 1;
    for (int i = 0; i < N; ++i) {
        const size_t q = (size_t)a * N + i;
        if (c.done_mode == ORC_DONE_XLESS) {
            if (e.agent_done[q] == 0 && ar.pos[i].x < c.done_x_thresh) {
                e.agent_done[q] = 1;
                e.arrive_step[q] = e.step_count[a];
                e.goal_x[q] = e.goal2_x[q];
                e.goal_y[q] = e.goal2_y[q];
                e.st[a].goals_reached += 1;
            }
        } else {
            const double dx = (double)ar.pos[i].x - e.goal_x[q];
            const double dy = (double)ar.pos[i].y - e.goal_y[q];
            const double lim = 2.0 * (double)c.radius;
            const bool reached = (dx * dx + dy * dy) < lim * lim; /* ALAN:555 */
            if (c.done_mode == ORC_DONE_GOAL) {
                if (e.agent_done[q] == 0 && reached) {
                    e.agent_done[q] = 1;
                    e.arrive_step[q] = e.step_count[a];
                    e.goal_x[q] = e.goal2_x[q];
                    e.goal_y[q] = e.goal2_y[q];
                    e.st[a].goals_reached += 1;
                }
            } else if (reached) {
                double u0, u1;
                rng2(c.seed, c.arena_offset + a, i, RNG_REGOAL, (uint32_t)e.regoal_count[q], &u0, &u1);
                e.goal_x[q] = uniform64((double)c.goal_x0, (double)c.goal_x1, u0);
                e.goal_y[q] = uniform64((double)c.goal_y0, (double)c.goal_y1, u1);
                e.regoal_count[q] += 1;
                e.st[a].goals_reached += 1;
            }
        }
        if (e.agent_done[q] == 0) all = 0;
    }
    return all;
}

/* env.py:461-488 for one arena: new positions only; velocities, targets and neighbour lists stay */
void arena_reset(Env& e, int a, const float* px, const float* py) {
    Arena& ar = e.arenas[a];
    const int N = e.N();
    const orc_config& c = e.cfg;
    for (int i = 0; i < N; ++i) {
        const size_t q = (size_t)a * N + i;
        if (px) {
            ar.pos[i] = mk(px[q], py[q]);
        } else {
            double u0, u1;
            rng2(c.seed, c.arena_offset + a, i, RNG_RESET, (uint32_t)e.episode[a], &u0, &u1);
            ar.pos[i] = mk((float)uniform64((double)c.spawn_x0, (double)c.spawn_x1, u0),
                           (float)uniform64((double)c.spawn_y0, (double)c.spawn_y1, u1));
        }
        e.agent_done[q] = 0;
    }
    set_pref_toward_goal(e, a);
    e.step_count[a] = 0;
    e.episode[a] += 1;
}

/* bookkeeping for callers that auto-reset: length of the episode that just ended and how many agents arrived */
void note_episode_end(Env& e, int a) {
    uint64_t arrived = 0;
    for (int i = 0; i < e.N(); ++i) arrived += e.agent_done[(size_t)a * e.N() + i] != 0;
    e.last_episode[a] = ((uint64_t)(uint32_t)e.step_count[a] << 32) | arrived;
}

template <class T>
void arena_reward(Env& e, int a) { /* env.py:389-400 */
    Arena& ar = e.arenas[a];
    const int N = e.N();
    const T scale = (T)e.cfg.reward_scale;
    for (int i = 0; i < N; ++i) {
        const size_t q = (size_t)a * N + i;
        const T vx = (T)ar.vel[i].x, vy = (T)ar.vel[i].y;
        const T gx = (T)e.pf64[2 * q], gy = (T)e.pf64[2 * q + 1];
        const T lx = (T)e.rl64[2 * q], ly = (T)e.rl64[2 * q + 1];
        const T r_goal = vx * gx + vy * gy;
        const T r_polite = vx * lx + vy * ly;
        const T r = scale * r_goal + (T(1) - scale) * r_polite;
        e.reward64[q] = (double)r;
        e.reward[q] = (float)r;
        e.st[a].sum_reward += (double)r;
    }
}

/* one arena through one env step.  actions != null: env.py:367-416 `step`;
 * actions == null: env.py:447-450 / ALAN:631-636 `orca_step` (+ the caller's done test) */
void arena_step(Env& e, int a, const float* actions, uint32_t flags, int prec) {
    if ((flags & ORC_F_FREEZE) && e.arena_done[a]) { e.frozen_steps[a] += 1; return; }
    Arena& ar = e.arenas[a];
    const int N = e.N();
    const orc_config& c = e.cfg;
    if (actions) {
        for (int i = 0; i < N; ++i) { /* env.py:371-383 */
            const size_t q = (size_t)a * N + i;
            double gx, gy, sn, cs;
            pref_dir64(ar.pos[i].x, ar.pos[i].y, e.goal_x[q], e.goal_y[q], &gx, &gy);
            sincos64((double)actions[q], &sn, &cs);
            /* (cos, sin)(atan2(g) + th) = rotation of g by th */
            const double lx = gx * cs - gy * sn, ly = gx * sn + gy * cs;
            e.pf64[2 * q] = gx; e.pf64[2 * q + 1] = gy;
            e.rl64[2 * q] = lx; e.rl64[2 * q + 1] = ly;
            ar.pref[i] = mk((float)lx, (float)ly);
        }
    }
    g_cur_arena = a;   /* (read by an armed capture only) */
    do_step(ar, e.tab(a), c.time_step, c.max_obst_neighbors, &e.st[a].obst_overflow);
    e.st[a].agent_steps += (uint64_t)N;
    if (flags & ORC_F_STATS) arena_collisions(e, a);
    if (actions) {
        if (prec == ORC_PREC_F64) arena_reward<double>(e, a);
        else arena_reward<float>(e, a);
    } else {
        set_pref_toward_goal(e, a); /* env.py:449 */
    }
    int all_done = 0;
    if (actions) { /* env.py:404-410: done_test, then the step counter */
        if (!(flags & ORC_F_NODONE)) all_done = arena_done_test(e, a);
        e.step_count[a] += 1;
    } else if (!(flags & ORC_F_NODONE)) { /* ALAN:118-121: counter, then done_test */
        e.step_count[a] += 1;
        all_done = arena_done_test(e, a);
    } /* env.py:447-458 orca_step: neither */
    if (c.max_step > 0 && e.step_count[a] >= c.max_step) all_done = 1;
    e.arena_done[a] = all_done;
    if (all_done) {
        e.st[a].episodes += 1;
        note_episode_end(e, a);
        if (flags & ORC_F_AUTORESET) arena_reset(e, a, nullptr, nullptr);
    }
    if (flags & ORC_F_OBS) arena_obs(e, a, prec);
}

/* ALAN_true.py:569-628 online_step for one arena (+ the step counter / done test of run_sim,
 * ALAN_true.py:119-121).  u: this arena's uniforms [N] or null. */
void arena_alan_step(Env& e, int a, const double* u, uint32_t flags, int prec) {
    if ((flags & ORC_F_FREEZE) && e.arena_done[a]) { e.frozen_steps[a] += 1; return; }
    Arena& ar = e.arenas[a];
    const int N = e.N(), nA = e.n_actions;
    const orc_config& c = e.cfg;
    std::vector<double> ps(nA), cdf(nA);
    for (int i = 0; i < N; ++i) {
        const size_t q = (size_t)a * N + i;
        double* w = &e.alan_w[q * nA];
        for (int k = 0; k < nA; ++k) ps[k] = exp64(w[k] / e.alan_temp);   /* :580-581 */
        const double sum = np_sum(ps.data(), nA);
        for (int k = 0; k < nA; ++k) ps[k] /= sum;                          /* :582 */
        /* np.random.choice(n, 1, p=ps): cdf = cumsum(p); cdf /= cdf[-1]; searchsorted(u, 'right') */
        double acc = 0.0;
        for (int k = 0; k < nA; ++k) { acc += ps[k]; cdf[k] = acc; }
        for (int k = 0; k < nA; ++k) cdf[k] /= acc;
        double ui;
        if (u) ui = u[i];
        else { /* a new stream every episode of the arena */
            double u1;
            rng2(c.seed, c.arena_offset + a, i, RNG_ALAN + (e.episode[a] << 8), (uint32_t)e.step_count[a], &ui, &u1);
        }
        int id = 0;
        while (id < nA - 1 && !(cdf[id] > ui)) ++id;
        e.alan_action[q] = id;
        double gx, gy;
        pref_dir64(ar.pos[i].x, ar.pos[i].y, e.goal_x[q], e.goal_y[q], &gx, &gy);   /* :588 */
        const double cs = e.act_c[id], sn = e.act_s[id];                             /* :592-595 */
        const double lx = gx * cs - gy * sn, ly = gx * sn + gy * cs;
        e.pf64[2 * q] = gx; e.pf64[2 * q + 1] = gy;
        e.rl64[2 * q] = lx; e.rl64[2 * q + 1] = ly;
        ar.pref[i] = mk((float)lx, (float)ly);                                        /* :598 */
    }
    g_cur_arena = a;   /* (read by an armed capture only) */
    do_step(ar, e.tab(a), c.time_step, c.max_obst_neighbors, &e.st[a].obst_overflow);  /* :601 */
    e.st[a].agent_steps += (uint64_t)N;
    if (flags & ORC_F_STATS) arena_collisions(e, a);
    if (prec == ORC_PREC_F64) arena_reward<double>(e, a);
    else arena_reward<float>(e, a);
    for (int i = 0; i < N; ++i) { /* :606-628: the weights are Python floats: always from the fp64 reward */
        const size_t q = (size_t)a * N + i;
        const double vx = (double)ar.vel[i].x, vy = (double)ar.vel[i].y;
        const double R = c.reward_scale * (vx * e.pf64[2 * q] + vy * e.pf64[2 * q + 1]) +
                         (1.0 - c.reward_scale) * (vx * e.rl64[2 * q] + vy * e.rl64[2 * q + 1]);
        double* w = &e.alan_w[q * nA];
        double* t = &e.alan_t[q * nA];
        for (int k = 0; k < nA; ++k) {
            t[k] += e.alan_dt;
            if (t[k] >= e.alan_window) { t[k] = 0.0; w[k] = 0.0; }
        }
        w[e.alan_action[q]] = R;
    }
    e.step_count[a] += 1;                                   /* ALAN:120 */
    int all_done = arena_done_test(e, a);                   /* ALAN:121 */
    if (c.max_step > 0 && e.step_count[a] >= c.max_step) all_done = 1;
    e.arena_done[a] = all_done;
    if (all_done) { e.st[a].episodes += 1; note_episode_end(e, a); }
    if (flags & ORC_F_OBS) arena_obs(e, a, prec);
}

/* ------------------------------------------------------------------------------------------ */
/* rvo2.PyRVOSimulator-shaped single simulator                                                 */
/* ------------------------------------------------------------------------------------------ */
struct Sim {
    float timeStep;
    AgentParams defaults;
    V2 defVel;
    Arena ar;
    std::vector<ObstVertex> obst;
};

template <class T>
int copy_out(const std::vector<T>& v, void* dst, size_t bytes) {
    if (bytes != v.size() * sizeof(T)) return -2;
    std::memcpy(dst, v.data(), bytes);
    return 0;
}
template <class T>
int copy_in(std::vector<T>& v, const void* src, size_t bytes) {
    if (bytes != v.size() * sizeof(T)) return -2;
    std::memcpy(v.data(), src, bytes);
    return 0;
}

}  // namespace

/* ============================================================================================ */
extern "C" {

void* orc_env_create(const orc_config* cfg) {
    if (!cfg || cfg->n_arenas <= 0 || cfg->n_agents <= 0 || cfg->max_neighbors < 0) return nullptr;
    Env* e = new Env();
    e->cfg = *cfg;
    const size_t A = cfg->n_arenas, N = cfg->n_agents;
    e->arenas.resize(A);
    AgentParams P{cfg->neighbor_dist, cfg->max_neighbors, cfg->time_horizon, cfg->time_horizon_obst,
                  cfg->radius, cfg->max_speed};
    for (size_t a = 0; a < A; ++a) {
        e->arenas[a].resize((int)N);
        for (size_t i = 0; i < N; ++i) {
            e->arenas[a].prm[i] = P;
            e->arenas[a].pos[i] = e->arenas[a].vel[i] = e->arenas[a].pref[i] = e->arenas[a].newVel[i] = mk(0, 0);
        }
    }
    e->goal_x.assign(A * N, 0); e->goal_y.assign(A * N, 0);
    e->goal2_x.assign(A * N, 0); e->goal2_y.assign(A * N, 0);
    e->reward.assign(A * N, 0); e->reward64.assign(A * N, 0);
    e->obs.assign(A * N * ORC_OBS_DIM, 0); e->obs64.assign(A * N * ORC_OBS_DIM, 0);
    e->obs_margin.assign(A * N * ORC_N_RAYS, std::numeric_limits<double>::infinity());
    e->agent_done.assign(A * N, 0); e->arrive_step.assign(A * N, -1); e->regoal_count.assign(A * N, 0);
    e->step_count.assign(A, 0); e->arena_done.assign(A, 0); e->episode.assign(A, 0);
    e->rl64.assign(A * N * 2, 0); e->pf64.assign(A * N * 2, 0);
    e->st.assign(A, orc_stats{});
    e->frozen_steps.assign(A, 0); e->last_episode.assign(A, 0);
    env_tables(*e);
    return e;
}

void orc_env_destroy(void* env) { delete (Env*)env; }

int orc_env_set_obstacles(void* env, const float* verts_xy, const int32_t* poly_sizes, int32_t n_poly) {
    Env* e = (Env*)env;
    e->obst.clear();
    e->obst_a.clear();
    for (Arena& ar : e->arenas) for (NbList& l : ar.obstNb) l.clear();  /* the old lists name edges of the old table */
    size_t off = 0;
    for (int p = 0; p < n_poly; ++p) {
        if (add_polygon(e->obst, verts_xy + 2 * off, poly_sizes[p]) < 0) return -1;
        off += poly_sizes[p];
    }
    process_obstacles(e->obst); /* env.py:123 */
    return 0;
}

/* a world per arena: n_poly[a] polygons for arena a, all polygons back to back (ALAN:359-372 per simulator) */
int orc_env_set_obstacles_per_arena(void* env, const float* verts_xy, const int32_t* poly_sizes, const int32_t* n_poly) {
    Env* e = (Env*)env;
    e->obst.clear();
    for (Arena& ar : e->arenas) for (NbList& l : ar.obstNb) l.clear();  /* the old lists name edges of the old tables */
    e->obst_a.assign(e->A(), std::vector<ObstVertex>());
    size_t voff = 0, poff = 0;
    for (int a = 0; a < e->A(); ++a) {
        for (int p = 0; p < n_poly[a]; ++p) {
            if (add_polygon(e->obst_a[a], verts_xy + 2 * voff, poly_sizes[poff]) < 0) return -1;
            voff += poly_sizes[poff];
            ++poff;
        }
        process_obstacles(e->obst_a[a]);
    }
    return 0;
}

static int table_out(const std::vector<ObstVertex>& ob, float* px, float* py, float* ux, float* uy, int32_t* next,
                     int32_t* prev, int32_t* convex, int32_t cap) {
    const int n = (int)ob.size();
    for (int i = 0; i < n && i < cap; ++i) {
        px[i] = ob[i].p.x; py[i] = ob[i].p.y;
        ux[i] = ob[i].unitDir.x; uy[i] = ob[i].unitDir.y;
        next[i] = ob[i].next; prev[i] = ob[i].prev; convex[i] = ob[i].convex ? 1 : 0;
    }
    return n;
}
int orc_env_obstacle_table(void* env, float* px, float* py, float* ux, float* uy, int32_t* next,
                           int32_t* prev, int32_t* convex, int32_t cap) {
    return table_out(((Env*)env)->tab(0), px, py, ux, uy, next, prev, convex, cap);
}
int orc_env_obstacle_table_arena(void* env, int32_t arena, float* px, float* py, float* ux, float* uy, int32_t* next,
                                 int32_t* prev, int32_t* convex, int32_t cap) {
    return table_out(((Env*)env)->tab(arena), px, py, ux, uy, next, prev, convex, cap);
}

/* scenario generators: agents only (obstacles come through orc_env_set_obstacles) */
int orc_env_init_scenario(void* env, int32_t scenario) {
    Env* e = (Env*)env;
    const orc_config& c = e->cfg;
    const int A = e->A(), N = e->N();
    const double r = (double)c.radius;
    for (int a = 0; a < A; ++a) {
        Arena& ar = e->arenas[a];
        const int64_t g = c.arena_offset + a;
        double theta = 0.0;
        for (int i = 0; i < N; ++i) {
            const size_t q = (size_t)a * N + i;
            double u0, u1, s, cs;
            rng2(c.seed, g, i, RNG_HEADING, 0, &u0, &u1);
            sincos64(uniform64(0.0, 2.0 * M_PI, u0), &s, &cs); /* env.py:89-90, ALAN:276-277 */
            ar.vel[i] = mk((float)cs, (float)s);
            if (scenario == ORC_SCN_CROWD || scenario == ORC_SCN_CROWD_SEPARATED) { /* ALAN:270-283 */
                const double E = std::sqrt(2.0 * r * N) * 2.0;
                /* SURVEY 8d rejection-sampled variant: redraw (sequence 1, 2, ... <= 63) while the start is
                   closer than 2 r to the start of an earlier agent */
                const int tries = scenario == ORC_SCN_CROWD_SEPARATED ? 64 : 1;
                const float minSq = sqr(c.radius + c.radius);
                for (int t = 0; t < tries; ++t) {
                    rng2(c.seed, g, i, RNG_POS, (uint32_t)t, &u0, &u1);
                    ar.pos[i] = mk((float)uniform64(0.0, E, u0), (float)uniform64(0.0, E, u1));
                    bool clear = true;
                    for (int j = 0; j < i && clear; ++j) clear = !(absSq(ar.pos[i] - ar.pos[j]) < minSq);
                    if (clear) break;
                }
                rng2(c.seed, g, i, RNG_GOAL, 0, &u0, &u1);
                e->goal_x[q] = uniform64(0.0, E, u0);
                e->goal_y[q] = uniform64(0.0, E, u1);
                e->goal2_x[q] = e->goal_x[q];
                e->goal2_y[q] = e->goal_y[q];
            } else if (scenario == ORC_SCN_CIRCLE) { /* ALAN:297-322 */
                const double circ = r * 3 * N;
                const double R = circ / (2.0 * M_PI);
                const double E = 2.0 * R + 4.0 * r;
                ar.pos[i] = mk((float)(E / 2 + R * std::cos(theta)), (float)(E / 2 + R * std::sin(theta)));
                e->goal_x[q] = E / 2 + R * std::cos(theta + M_PI);
                e->goal_y[q] = E / 2 + R * std::sin(theta + M_PI);
                e->goal2_x[q] = e->goal_x[q];
                e->goal2_y[q] = e->goal_y[q];
                theta += (2.0 * M_PI) / N;
            } else if (scenario == ORC_SCN_DOORWAY) { /* env.py:86-95 */
                const double E = 10.0;
                rng2(c.seed, g, i, RNG_POS, 0, &u0, &u1);
                ar.pos[i] = mk((float)uniform64(E * 0.5, E, u0), (float)uniform64(0.0, E, u1));
                e->goal_x[q] = 1.0; e->goal_y[q] = 5.0;
                e->goal2_x[q] = -10.0; e->goal2_y[q] = 5.0; /* env.py:361 */
            } else if (scenario >= 3 && scenario <= 6) { /* ALAN:175-193, 213-258, 333-357, 377-416 */
                const double E = (scenario == 3) ? std::sqrt(2 * r * N) * 3
                               : (scenario == 5) ? 3 * r * N : std::sqrt(2 * r * N) * 10;
                double px = 0, py = 0, gx = 0, gy = 0, g2x = 0, g2y = 0;
                if (scenario == 3) { /* congested */
                    rng2(c.seed, g, i, RNG_POS, 0, &u0, &u1);
                    px = uniform64(E * 0.2, E, u0); py = uniform64(0.0, E, u1);
                    gx = 0.1 * E - 1.0; gy = E / 2; g2x = 0.1 * E - E; g2y = E / 2;
                } else if (scenario == 4) { /* incoming */
                    if (i == 0) {
                        px = 0.1 * E; py = E / 2; gx = g2x = 0.9 * E; gy = g2y = E / 2;
                    } else {
                        const double len = std::sqrt((double)(N - 1)), x_inc = 3 * r, y_inc = 2.1 * r;
                        const double y_start = E / 2 - ((y_inc * len) / 2);
                        double x_pos = 0.8 * E, y_pos = y_start;
                        for (int k = 1; k < i; ++k) { /* replay the placement loop up to this agent */
                            y_pos += y_inc;
                            if (y_pos > y_start + y_inc * len) { x_pos += x_inc; y_pos = y_start; }
                        }
                        px = x_pos; py = y_pos; gx = g2x = x_pos - 0.7 * E; gy = g2y = y_pos;
                    }
                } else if (scenario == 5) { /* blocks */
                    double y_pos = 1.5 * r;
                    for (int k = 0; k < i; ++k) y_pos += 3 * r;
                    px = 1.5 * r; py = y_pos; gx = g2x = E - 1.5 * r; gy = g2y = y_pos;
                } else { /* deadlock */
                    const int half = N / 2;
                    if (i < half) {
                        double pos_x = 0.2 * E;
                        for (int k = 0; k < i; ++k) pos_x += -3 * r;
                        px = pos_x; gx = 0.9 * E; g2x = 0.9 * E + E;
                    } else {
                        double pos_x = 0.8 * E;
                        for (int k = half; k < i; ++k) pos_x += 3 * r;
                        px = pos_x; gx = 0.1 * E; g2x = 0.1 * E - E;
                    }
                    py = E / 2; gy = g2y = E / 2;
                }
                ar.pos[i] = mk((float)px, (float)py);
                e->goal_x[q] = gx; e->goal_y[q] = gy;
                e->goal2_x[q] = g2x; e->goal2_y[q] = g2y;
            } else {
                return -1;
            }
            e->agent_done[q] = 0; e->arrive_step[q] = -1; e->regoal_count[q] = 0;
            ar.agentNb[i].clear(); ar.obstNb[i].clear();
        }
        set_pref_toward_goal(*e, a); /* env.py:97 */
        e->step_count[a] = 0; e->arena_done[a] = 0; e->episode[a] = 0;
    }
    return 0;
}

#define ARENA_FIELD(member, comp)                                                        \
    {                                                                                    \
        std::vector<float> tmp((size_t)A * N);                                           \
        if (write) {                                                                     \
            if (bytes != tmp.size() * 4) return -2;                                      \
            std::memcpy(tmp.data(), src, bytes);                                         \
            for (int a = 0; a < A; ++a)                                                  \
                for (int i = 0; i < N; ++i) e->arenas[a].member[i].comp = tmp[(size_t)a * N + i]; \
            return 0;                                                                    \
        }                                                                                \
        for (int a = 0; a < A; ++a)                                                      \
            for (int i = 0; i < N; ++i) tmp[(size_t)a * N + i] = e->arenas[a].member[i].comp; \
        return copy_out(tmp, dst, bytes);                                                \
    }

static int env_access(Env* e, int field, const void* src, void* dst, size_t bytes, bool write) {
    const int A = e->A(), N = e->N();
    switch (field) {
        case ORC_FLD_POS_X: ARENA_FIELD(pos, x)
        case ORC_FLD_POS_Y: ARENA_FIELD(pos, y)
        case ORC_FLD_VEL_X: ARENA_FIELD(vel, x)
        case ORC_FLD_VEL_Y: ARENA_FIELD(vel, y)
        case ORC_FLD_PREF_X: ARENA_FIELD(pref, x)
        case ORC_FLD_PREF_Y: ARENA_FIELD(pref, y)
        case ORC_FLD_GOAL_X: return write ? copy_in(e->goal_x, src, bytes) : copy_out(e->goal_x, dst, bytes);
        case ORC_FLD_GOAL_Y: return write ? copy_in(e->goal_y, src, bytes) : copy_out(e->goal_y, dst, bytes);
        case ORC_FLD_GOAL2_X: return write ? copy_in(e->goal2_x, src, bytes) : copy_out(e->goal2_x, dst, bytes);
        case ORC_FLD_GOAL2_Y: return write ? copy_in(e->goal2_y, src, bytes) : copy_out(e->goal2_y, dst, bytes);
        case ORC_FLD_REWARD: return write ? -3 : copy_out(e->reward, dst, bytes);
        case ORC_FLD_REWARD64: return write ? -3 : copy_out(e->reward64, dst, bytes);
        case ORC_FLD_OBS: return write ? -3 : copy_out(e->obs, dst, bytes);
        case ORC_FLD_OBS64: return write ? -3 : copy_out(e->obs64, dst, bytes);
        case ORC_FLD_OBS_MARGIN: return write ? -3 : copy_out(e->obs_margin, dst, bytes);
        case ORC_FLD_AGENT_DONE: return write ? copy_in(e->agent_done, src, bytes) : copy_out(e->agent_done, dst, bytes);
        case ORC_FLD_ARRIVE_STEP: return write ? copy_in(e->arrive_step, src, bytes) : copy_out(e->arrive_step, dst, bytes);
        case ORC_FLD_REGOAL_COUNT: return write ? copy_in(e->regoal_count, src, bytes) : copy_out(e->regoal_count, dst, bytes);
        case ORC_FLD_STEP_COUNT: return write ? copy_in(e->step_count, src, bytes) : copy_out(e->step_count, dst, bytes);
        case ORC_FLD_ARENA_DONE: return write ? copy_in(e->arena_done, src, bytes) : copy_out(e->arena_done, dst, bytes);
        case ORC_FLD_EPISODE: return write ? copy_in(e->episode, src, bytes) : copy_out(e->episode, dst, bytes);
        case ORC_FLD_ALAN_WEIGHTS: return write ? copy_in(e->alan_w, src, bytes) : copy_out(e->alan_w, dst, bytes);
        case ORC_FLD_ALAN_TIMES: return write ? copy_in(e->alan_t, src, bytes) : copy_out(e->alan_t, dst, bytes);
        case ORC_FLD_ALAN_ACTION: return write ? copy_in(e->alan_action, src, bytes) : copy_out(e->alan_action, dst, bytes);
        case ORC_FLD_ARENA_STATS: {
            if (write) return -3;
            std::vector<uint64_t> rows((size_t)e->A() * 8, 0);
            for (int a = 0; a < e->A(); ++a) {
                uint64_t* r = &rows[(size_t)a * 8];
                r[0] = e->st[a].episodes; r[1] = e->st[a].collisions; r[2] = e->st[a].obst_collisions;
                r[3] = e->st[a].goals_reached; r[4] = e->st[a].obst_overflow;
                std::memcpy(&r[5], &e->st[a].sum_reward, 8);
                r[6] = e->frozen_steps[a]; r[7] = e->last_episode[a];
            }
            return copy_out(rows, dst, bytes);
        }
        case ORC_FLD_NB_COUNT:
        case ORC_FLD_OBST_COUNT: {
            std::vector<int32_t> tmp((size_t)A * N);
            if (write) {
                /* restoring counts alone is meaningless; lists are restored through *_IDX */
                return -3;
            }
            for (int a = 0; a < A; ++a)
                for (int i = 0; i < N; ++i)
                    tmp[(size_t)a * N + i] = (int32_t)(field == ORC_FLD_NB_COUNT ? e->arenas[a].agentNb[i].size()
                                                                                  : e->arenas[a].obstNb[i].size());
            return copy_out(tmp, dst, bytes);
        }
        case ORC_FLD_NB_IDX:
        case ORC_FLD_OBST_IDX: {
            /* [A,N,W] i32, padded with -1; writing restores the lists (distances are not kept:
             * they are only used while a list is being built) */
            const int W = field == ORC_FLD_NB_IDX ? e->cfg.max_neighbors : e->cfg.max_obst_neighbors;
            std::vector<int32_t> tmp((size_t)A * N * W, -1);
            if (write) {
                if (bytes != tmp.size() * 4) return -2;
                std::memcpy(tmp.data(), src, bytes);
            }
            for (int a = 0; a < A; ++a)
                for (int i = 0; i < N; ++i) {
                    NbList& l = field == ORC_FLD_NB_IDX ? e->arenas[a].agentNb[i] : e->arenas[a].obstNb[i];
                    int32_t* row = &tmp[((size_t)a * N + i) * W];
                    if (write) {
                        l.clear();
                        for (int k = 0; k < W && row[k] >= 0; ++k) l.push_back(std::make_pair(0.0f, (int)row[k]));
                    } else {
                        for (int k = 0; k < W && k < (int)l.size(); ++k) row[k] = l[k].second;
                    }
                }
            return write ? 0 : copy_out(tmp, dst, bytes);
        }
        default: return -1;
    }
}

int orc_env_set(void* env, int32_t field, const void* src, size_t bytes) {
    return env_access((Env*)env, field, src, nullptr, bytes, true);
}
int orc_env_get(void* env, int32_t field, void* dst, size_t bytes) {
    return env_access((Env*)env, field, nullptr, dst, bytes, false);
}

int orc_env_reset(void* env, const float* pos_x, const float* pos_y, uint32_t flags, int32_t prec) {
    Env* e = (Env*)env;
    for (int a = 0; a < e->A(); ++a) {
        arena_reset(*e, a, pos_x, pos_y);
        e->arena_done[a] = 0;
        if (flags & ORC_F_OBS) arena_obs(*e, a, prec);
    }
    return 0;
}

int orc_env_reset_masked(void* env, const int32_t* mask, uint32_t flags, int32_t prec) {
    Env* e = (Env*)env;
    if (!mask) return -1;
    for (int a = 0; a < e->A(); ++a) {
        if (!mask[a]) continue;
        arena_reset(*e, a, nullptr, nullptr);
        e->arena_done[a] = 0;
        if (flags & ORC_F_OBS) arena_obs(*e, a, prec);
    }
    return 0;
}

int orc_env_step(void* env, const float* actions, uint32_t flags, int32_t prec) {
    Env* e = (Env*)env;
    if (!actions) return -1;
    for (int a = 0; a < e->A(); ++a) arena_step(*e, a, actions, flags, prec);
    return 0;
}

int orc_env_orca_step(void* env, uint32_t flags, int32_t prec) {
    Env* e = (Env*)env;
    for (int a = 0; a < e->A(); ++a) arena_step(*e, a, nullptr, flags, prec);
    return 0;
}

/* one env step with the arenas dealt to n_threads threads (arenas never interact): the multi-core CPU baseline */
int orc_env_step_mt(void* env, const float* actions, uint32_t flags, int32_t prec, int32_t n_threads) {
    Env* e = (Env*)env;
    const int A = e->A();
    if (n_threads <= 1) {
        for (int a = 0; a < A; ++a) arena_step(*e, a, actions, flags, prec);
        return 0;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([=]() { for (int a = t; a < A; a += n_threads) arena_step(*e, a, actions, flags, prec); });
    for (auto& t : th) t.join();
    return 0;
}

int orc_env_rollout(void* env, int32_t steps, uint32_t flags, int32_t n_threads) {
    Env* e = (Env*)env;
    const int A = e->A();
    if (n_threads <= 1) {
        for (int a = 0; a < A; ++a)
            for (int s = 0; s < steps; ++s) arena_step(*e, a, nullptr, flags, ORC_PREC_F32);
        return 0;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) {
        th.emplace_back([=]() {
            for (int a = t; a < A; a += n_threads)
                for (int s = 0; s < steps; ++s) arena_step(*e, a, nullptr, flags, ORC_PREC_F32);
        });
    }
    for (auto& t : th) t.join();
    return 0;
}

/* The multi-core CPU baseline: `steps` env steps with the arenas dealt in contiguous blocks to n_threads threads, each
 * thread stepping ITS arenas through the whole sample (arena-major: an arena's state stays in that core's cache) with no
 * per-step hand-off or join -- arenas never interact, so nothing has to meet between steps.  actions_pool: null
 * (ORCA-only step) or [pool, A, N] heading offsets, step s uses pool entry s % pool (the same action tensor for every
 * arena-step that orc_env_step_mt would have been handed step by step, so the two give identical states). */
int orc_env_rollout_mt(void* env, const float* actions_pool, int32_t pool, int32_t steps, uint32_t flags, int32_t prec,
                       int32_t n_threads) {
    Env* e = (Env*)env;
    const int A = e->A();
    const size_t an = (size_t)A * e->N();
    if (actions_pool && pool < 1) return -1;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > A) n_threads = A;
    auto work = [=](int t) {
        const int a0 = (int)((int64_t)A * t / n_threads), a1 = (int)((int64_t)A * (t + 1) / n_threads);
        for (int a = a0; a < a1; ++a)
            for (int s = 0; s < steps; ++s)
                arena_step(*e, a, actions_pool ? actions_pool + (size_t)(s % pool) * an : nullptr, flags, prec);
    };
    if (n_threads == 1) { work(0); return 0; }
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
    for (auto& t : th) t.join();
    return 0;
}

int orc_env_alan_configure(void* env, const double* actions_xy, int32_t n_actions, double temp, double timewindow,
                           double time_step) {
    Env* e = (Env*)env;
    if (n_actions < 1 || n_actions > 32 || !actions_xy) return -1;
    e->n_actions = n_actions;
    e->alan_temp = temp;
    e->alan_window = timewindow;
    e->alan_dt = time_step;
    e->act_c.resize(n_actions); e->act_s.resize(n_actions);
    for (int k = 0; k < n_actions; ++k) { /* (cos, sin) of atan2(action) = the normalised action */
        const double x = actions_xy[2 * k], y = actions_xy[2 * k + 1], len = std::sqrt(x * x + y * y);
        e->act_c[k] = len == 0.0 ? 1.0 : x / len;
        e->act_s[k] = len == 0.0 ? 0.0 : y / len;
    }
    const size_t an = (size_t)e->A() * e->N();
    e->alan_w.assign(an * n_actions, 0.0);
    e->alan_t.assign(an * n_actions, 0.0);
    e->alan_action.assign(an, 0);
    return 0;
}

int orc_env_alan_step(void* env, const double* u, uint32_t flags, int32_t prec) {
    Env* e = (Env*)env;
    if (e->n_actions <= 0) return -1;
    for (int a = 0; a < e->A(); ++a) arena_alan_step(*e, a, u ? u + (size_t)a * e->N() : nullptr, flags, prec);
    return 0;
}

int orc_env_stats(void* env, orc_stats* out) {
    Env* e = (Env*)env;
    orc_stats s{};
    for (int a = 0; a < e->A(); ++a) {
        s.agent_steps += e->st[a].agent_steps; s.episodes += e->st[a].episodes;
        s.collisions += e->st[a].collisions; s.obst_collisions += e->st[a].obst_collisions;
        s.goals_reached += e->st[a].goals_reached; s.obst_overflow += e->st[a].obst_overflow;
        s.sum_reward += e->st[a].sum_reward;
    }
    *out = s;
    return 0;
}

/* ---------------- single simulator ---------------- */
void* orc_sim_create(float time_step, float neighbor_dist, int32_t max_neighbors, float time_horizon,
                     float time_horizon_obst, float radius, float max_speed, float vx, float vy) {
    Sim* s = new Sim();
    s->timeStep = time_step;
    s->defaults = AgentParams{neighbor_dist, max_neighbors, time_horizon, time_horizon_obst, radius, max_speed};
    s->defVel = mk(vx, vy);
    return s;
}
void orc_sim_destroy(void* sim) { delete (Sim*)sim; }
int orc_sim_add_agent(void* sim, float x, float y, float neighbor_dist, int32_t max_neighbors,
                      float time_horizon, float time_horizon_obst, float radius, float max_speed,
                      float vx, float vy) {
    Sim* s = (Sim*)sim;
    const int i = s->ar.n();
    s->ar.resize(i + 1);
    s->ar.pos[i] = mk(x, y);
    s->ar.vel[i] = mk(vx, vy);
    s->ar.pref[i] = mk(0, 0);
    s->ar.newVel[i] = mk(0, 0);
    s->ar.prm[i] = AgentParams{neighbor_dist, max_neighbors, time_horizon, time_horizon_obst, radius, max_speed};
    return i;
}
int orc_sim_add_obstacle(void* sim, const float* xy, int32_t n) { return add_polygon(((Sim*)sim)->obst, xy, n); }
void orc_sim_process_obstacles(void* sim) { process_obstacles(((Sim*)sim)->obst); }
void orc_sim_do_step(void* sim) {
    Sim* s = (Sim*)sim;
    g_cur_arena = 0;   /* (a simulator is one arena: orc_debug_capture(0, agent)) */
    do_step(s->ar, s->obst, s->timeStep, std::numeric_limits<int>::max(), nullptr);
}
int orc_sim_num_agents(void* sim) { return ((Sim*)sim)->ar.n(); }
void orc_sim_get_agent(void* sim, int32_t i, int32_t what, float* out2) {
    Sim* s = (Sim*)sim;
    const V2 v = what == 0 ? s->ar.pos[i] : what == 1 ? s->ar.vel[i] : s->ar.pref[i];
    out2[0] = v.x; out2[1] = v.y;
}
void orc_sim_set_agent(void* sim, int32_t i, int32_t what, float x, float y) {
    Sim* s = (Sim*)sim;
    (what == 0 ? s->ar.pos[i] : what == 1 ? s->ar.vel[i] : s->ar.pref[i]) = mk(x, y);
}
int orc_sim_num_agent_neighbors(void* sim, int32_t i) { return (int)((Sim*)sim)->ar.agentNb[i].size(); }
int orc_sim_agent_neighbor(void* sim, int32_t i, int32_t k) { return ((Sim*)sim)->ar.agentNb[i][k].second; }
int orc_sim_num_obstacle_neighbors(void* sim, int32_t i) { return (int)((Sim*)sim)->ar.obstNb[i].size(); }
int orc_sim_obstacle_neighbor(void* sim, int32_t i, int32_t k) { return ((Sim*)sim)->ar.obstNb[i][k].second; }
int orc_sim_next_obstacle_vertex(void* sim, int32_t v) { return ((Sim*)sim)->obst[v].next; }
int orc_sim_prev_obstacle_vertex(void* sim, int32_t v) { return ((Sim*)sim)->obst[v].prev; }
void orc_sim_obstacle_vertex(void* sim, int32_t v, float* out2) {
    out2[0] = ((Sim*)sim)->obst[v].p.x; out2[1] = ((Sim*)sim)->obst[v].p.y;
}
int orc_sim_num_obstacle_vertices(void* sim) { return (int)((Sim*)sim)->obst.size(); }

/* ---------------- stand-alone pieces ---------------- */
int orc_line_intersection_f64(const double* L1, const double* L2, double* d, double* ix, double* iy) {
    /* general p0 (the reference's rays start at the origin; tests also probe p0 != 0) */
    const double p0x = L1[0], p0y = L1[1], p1x = L1[2], p1y = L1[3];
    const double p2x = L2[0], p2y = L2[1], p3x = L2[2], p3y = L2[3];
    const double s10x = p1x - p0x, s10y = p1y - p0y, s32x = p3x - p2x, s32y = p3y - p2y;
    const double denom = s10x * s32y - s32x * s10y;
    *d = std::numeric_limits<double>::infinity(); *ix = 0; *iy = 0;
    if (denom == 0) return 0;
    const bool dpos = denom > 0;
    const double s02x = p0x - p2x, s02y = p0y - p2y;
    const double s_numer = s10x * s02y - s10y * s02x;
    if ((s_numer < 0) == dpos) return 0;
    const double t_numer = s32x * s02y - s32y * s02x;
    if ((t_numer < 0) == dpos) return 0;
    if (((s_numer > denom) == dpos) || ((t_numer > denom) == dpos)) return 0;
    const double t = t_numer / denom;
    *ix = p0x + t * s10x; *iy = p0y + t * s10y;
    *d = std::sqrt(*ix * *ix + *iy * *iy);
    return 1;
}
void orc_comp_laser_f64(const double* ray_ends, const double* segs, int32_t m, const double* o, double* out) {
    double ox, oy; /* normalise like comp_pref_vel would have */
    const double len = std::sqrt(o[0] * o[0] + o[1] * o[1]);
    if (len == 0) { ox = 1; oy = 0; } else { ox = o[0] / len; oy = o[1] / len; }
    comp_laser<double>(ray_ends, (const Seg<double>*)(const void*)segs, m, ox, -oy, out);
}
void orc_comp_laser_f32(const float* ray_ends, const float* segs, int32_t m, const float* o, float* out) {
    double ox, oy;
    const double len = std::sqrt((double)o[0] * o[0] + (double)o[1] * o[1]);
    if (len == 0) { ox = 1; oy = 0; } else { ox = o[0] / len; oy = o[1] / len; }
    comp_laser<float>(ray_ends, (const Seg<float>*)(const void*)segs, m, (float)ox, (float)(-oy), out);
}
void orc_debug_counters(uint64_t* out4) { for (int i = 0; i < 4; ++i) out4[i] = g_dbg[i]; }
/* branch counters of the CALLING thread (serial stepping: tests), names in the same order */
int orc_debug_branch_count(void) { return BR__COUNT; }
const char* orc_debug_branch_name(int k) { return (k >= 0 && k < BR__COUNT) ? g_br_names[k] : ""; }
void orc_debug_branches(uint64_t* out, int n) { for (int i = 0; i < n && i < BR__COUNT; ++i) out[i] = g_br[i]; }
void orc_debug_branches_reset(void) { for (int i = 0; i < BR__COUNT; ++i) g_br[i] = 0; }
/* capture the ORCA lines of agent `agent` of arena `arena` (local index) during the NEXT serial step on this thread */
void orc_debug_capture(int arena, int agent) { g_cap.arena = arena; g_cap.agent = agent; g_cap.armed = (arena >= 0) ? 1 : 0; }
/* -> number of lines (obstacle lines first), or -1 if nothing was captured.  lines4: [cap][4] = point.x, point.y, dir.x, dir.y;
 * line_edge: [cap] edge id of each OBSTACLE line; nb_tag: [cap] per obstacle neighbour, edge id << 8 | branch index;
 * info4: numObstLines, obstacle neighbours, the line LP2 failed at (= lines: feasible), lines */
int orc_debug_captured(float* lines4, int* line_edge, int* nb_tag, int cap, int* info4) {
    if (g_cap.armed != 2) return -1;
    for (int k = 0; k < g_cap.nl && k < cap; ++k) {
        lines4[4 * k] = g_cap.lines[k].point.x; lines4[4 * k + 1] = g_cap.lines[k].point.y;
        lines4[4 * k + 2] = g_cap.lines[k].dir.x; lines4[4 * k + 3] = g_cap.lines[k].dir.y;
    }
    for (int k = 0; k < (int)g_cap.edge.size() && k < cap; ++k) line_edge[k] = g_cap.edge[k];
    for (int k = 0; k < (int)g_cap.tag.size() && k < cap; ++k) nb_tag[k] = g_cap.tag[k];
    info4[0] = g_cap.numObst; info4[1] = (int)g_cap.tag.size(); info4[2] = g_cap.fail; info4[3] = g_cap.nl;
    return g_cap.nl;
}
void orc_sincos64(double a, double* s, double* c) { sincos64(a, s, c); }
double orc_exp64(double x) { return exp64(x); }
void orc_pref_dir64(float px, float py, double gx, double gy, double* out2) { pref_dir64(px, py, gx, gy, &out2[0], &out2[1]); }
void orc_philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out4) {
    philox4x32(c0, c1, c2, c3, k0, k1, out4);
}
void orc_ray_table(double nd, double* out32) { ray_table(nd, out32); }
void orc_octagon_table(double r, double* out32) { octagon_table(r, out32); }

} /* extern "C" */
