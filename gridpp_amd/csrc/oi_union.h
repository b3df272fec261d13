// k_oi_union: optimal interpolation with ONE factorisation per tile of 64 grid cells.
//
// The 64 cells of a tile select almost the same observations (src/api/oi.cpp:229-273): on the headline workload the
// union U of the 64 selections has ~33 members of which ~28 (the core C) are selected by every cell.  With the core
// ordered first, the Cholesky factor of (P+R)[S,S] for a cell's selection S = C + E_cell is
//     L_S = | L_C   0  |      B = (P+R)[E,C] L_C^-T,   L_E L_E^T = (P+R)[E,E] - B B^T   (Schur complement)
//           | B    L_E |
// so the wave factorises the core once (rows in lanes, extras and obs-background as extra rows, which leaves B, the
// Schur complement of ALL extras and L_C^-1 d behind), and then every LANE finishes its own cell: a forward substitution
// of its G vector (the rho values kept from the scan) against L_C read from LDS, and a tiny Cholesky (<= 6 rows) on the
// sub-block of the Schur complement that belongs to its own extras.  increment = (L^-1 g).(L^-1 d) and
// 1 - K G^T = 1 - |L^-1 g|^2 as in k_oi (oi.cpp:315-316,336).
//
// A tile that does not fit (more than 40 live candidates during the scan, union > 40, more than 12 extras or more than
// 6 per cell) is appended to a work list: it is retried as four 16-cell items, those as 4-cell items, and what is still
// left is done by k_oi (one factorisation per distinct selection).
#pragma once
#include "oi_common.h"

#pragma clang fp contract(off)

struct OiArgs {
    const float *gx, *gy, *gz, *gelev, *glaf, *bg, *bvar;
    float *out, *out_var;
    int C, ny, nx, tiles_x, ntiles, tiled2d;
    int wshift;              // log2 of the tile width in columns (tile = (64 >> wshift) rows x (1 << wshift) columns)
    ScanArgs s;
    const float4* ogeo;      // original order
    const float4* oaux;      // original order: laf, obs, pbg, ratio
    const float4* saux;      // the same records at the sorted positions (k_oi_union: one dependent load less per tile)
    int S, allow_extrap;
    int* err;                // bit0: list overflow (needs the large-n path), bit1: singular / not SPD
    unsigned long long* counters;   // [80 + 2 k], [81 + 2 k]: cells updated, factorisations -- spread over GPP_NSLOT slots (k = block
                                    // index mod GPP_NSLOT): a quarter of a million atomics on ONE address would serialise in the L2
    // Work lists (device resident, lengths read on the device: no host round trip between the passes).
    //   pass 1  k_oi_union<., false> over all tiles                  -> out_list: tiles it declined
    //   pass 2  k_oi_union<., true>, level 1: 16-cell quarters        -> out_list: tile * 16 + quarter it declined
    //   pass 3  k_oi_union<., true>, level 2: 4-cell quarters of those -> out_list: tile * 16 + 4-cell item it declined
    //   pass 4  k_oi over that list (4 cells per wave); ~tile entries are whole tiles forwarded unsplit
    const int* in_list;      // NULL: all tiles
    const int* in_count;
    int* out_list;
    int* out_count;
    int nrun;                // tiles to run when in_list is NULL
    int tile0, tile_n;       // first pass of k_oi_union: the tiles [tile0, tile0 + tile_n) (all of them unless the host path runs the grid in bands of tile rows)
    int* tail_count;         // persistent first pass: counter of its dynamic tail (cleared with the status block)
    const unsigned char* skip_flags;   // first pass: tiles it leaves alone (NULL: none) -- the tiles this geometry declined in earlier calls, which the
                                       // list passes take from the remembered list on a second stream WHILE the first pass runs (oi.hip, `overlap`)
    int level;               // k_oi_union<., true>: 1 or 2 (see there)
    const int* parent_count; // level 2: length of the level-1 input list (how many tiles were split)
    int debug;               // GPP_OI_DEBUG: bit0 = skip the solve (timing experiments only)
    int pair_solo;           // k_oi_union_pair: never merge (A/B: what the two-wave workgroups cost by themselves)
    // k_oi -> k_oi_big: cells with more usable observations than the 62-row register tile holds
    int* big_list;           // cell indices
    int* big_count;
    unsigned long long* big_keys;   // per workgroup of k_oi_big: BIG_CAND candidate keys (sorted there)
    double* big_mat;         // per workgroup: (BIG_N + 2) x BIG_N matrix
    // k_oi_big -> k_oi_huge: cells beyond BIG_N / BIG_CAND, and every listed cell of a non-symmetric or spatially varying structure
    int* huge_list;
    int* huge_count;
    unsigned long long* huge_keys;  // per workgroup: huge_kcap candidate keys
    double* huge_mat;        // per workgroup: huge_ncap x (huge_ncap + 2) augmented matrix
    int huge_kcap, huge_ncap;
    // k_oi -> k_oi_pairs: the selections of cells that share their observation set with no other cell of their tile
    unsigned* pair_sel;      // [cell][32]: observation indices (original order)
    int* pair_n;             // [cell]: their number; 0 = the cell was finished by k_oi
};

#define GPP_NSLOT 512
#define BIG_CAND 8192   // candidates of one cell k_oi_big can sort
#define BIG_N 512       // observations of one cell k_oi_big can factorise
#define BIG_NL 104      // ... with the matrix in LDS (86 KB beside the 64 KB of candidate keys)
#define ERR_OVERFLOW 1
#define ERR_SINGULAR 2

// index (0..63) of the m-th set bit of mask (m < popcount(mask))
__device__ __forceinline__ int nth_set_bit(unsigned long long mask, int m) {
    int pos = 0;
#pragma unroll
    for(int w = 32; w > 0; w >>= 1) {
        unsigned long long lowmask = (w == 32) ? 0xffffffffull : ((1ull << w) - 1ull);
        int c = __popcll((mask >> pos) & lowmask);
        if(m >= c) { m -= c; pos += w; }
    }
    return pos;
}

#ifdef GPP_UNION_STATS     // diagnostic build: event counts in counters[4..12]
#define UNION_STATS true
#else
#define UNION_STATS false
#endif
#ifdef GPP_UNION_PROFILE   // diagnostic build: per-phase shader-clock totals in counters[20..]
#define UPROF(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if(lane == 0) L.prof[i] += (unsigned)(t_ - tprev); tprev = t_; } while(0)
#else
#define UPROF(i) do { } while(0)
#endif

// Two sizes of the kernel.  NC = 32: max_points <= 32, 32 register columns (+ 8 late columns kept in LDS), four waves per
// workgroup.  NC = 64: max_points 33..62, 64 register columns, two waves per workgroup (46 KB of LDS) at one wave per SIMD.
template <int NC> struct UnionCfg;
template <> struct UnionCfg<32> {
    static constexpr int WCAP = 40;     // candidate slots of a tile (live union during the scan)
    static constexpr int MAXU = 40;     // rows of the shared factorisation: 32 register columns + 8
    static constexpr int SOLVE = 1024;  // doubles of the shared-factor area
#ifndef GPP_UNION_PERSIST
// 0 (product): one tile per wave; 1: persistent grid, static striding; 2: persistent grid, the last eighth of the tiles from a counter.
// Measured on one box, round 4 (tools/ab_bench.sh, first pass of the headline): 0: 4.65 ms, 1: 5.31 ms, 2: 4.87 ms -- the tile loop costs 21
// registers of hoisted invariants (163 instead of 142) and scalar spills into vector lanes, which the end of the per-workgroup idling
// does not pay back; the hardware's workgroup dispatcher is the better load balancer here.
#define GPP_UNION_PERSIST 0
#endif
#ifndef GPP_UNION_WPB
#define GPP_UNION_WPB 2   // (round 6, after `late` went over the observation records: two waves per workgroup, six workgroups per CU, 4.36 ms per step against 4.41 with
                        //  four -- the LDS of a workgroup is released when its slower tile is done.  tools/ab_bench.sh, round 4: 3 waves per workgroup -- four workgroups per CU, the same twelve waves, finer release -- 4.52 ms per step
                        //  against 4.47; 6: 7.5 ms, one workgroup per CU)
#endif
    static constexpr int WPB = GPP_UNION_WPB;       // waves (work items) per workgroup
    template <bool PLAIN> static constexpr bool persistent() { return PLAIN && GPP_UNION_PERSIST != 0; }   // first pass as a persistent grid (see k_oi_union)
};
// NC = 48 (round 6): max_points 33 .. 46 -- 48 register columns + 8 late columns, 56 slots; 19 KB of LDS per wave and ~200 registers: TWO waves per SIMD
// where the 64-column form has 1.5 (its 22.8 KB per wave leave room for three two-wave workgroups per CU), and a quarter fewer multiply-adds.
template <> struct UnionCfg<48> {
    static constexpr int WCAP = 56;
    static constexpr int MAXU = 56;
    static constexpr int SOLVE = 1792;  // c = 46, 10 extras: 1756 doubles
    static constexpr int WPB = 2;
    template <bool PLAIN> static constexpr bool persistent() { return false; }
};
template <> struct UnionCfg<64> {
    static constexpr int WCAP = 64;
    static constexpr int MAXU = 62;     // lane 63 carries obs - background
    static constexpr int SOLVE = 2240;  // worst case c = 50, 12 extras: 2143 doubles, + the 64 of the column staging
    static constexpr int WPB = 2;
    template <bool PLAIN> static constexpr bool persistent() { return false; }
};
constexpr int U_MAXE = 12;     // union minus core
constexpr int U_MAXM = 6;      // extras of one cell

// Workgroup index -> position in the tile order, XCD-aware.  The hardware hands consecutive workgroups to the eight XCDs in turn (each XCD has its own
// L2), and with two tiles per workgroup a 128-byte line of a grid row (four 8-cell tiles) is shared by TWO consecutive workgroups: in launch order they sit
// on different XCDs and every line of the six cell arrays is fetched from HBM twice (FETCH_SIZE 379 MB per launch instead of 191: measured after
// GPP_UNION_WPB went from 4 to 2).  Inside every group of sixteen workgroups, workgroup b takes position 2 (b mod 8) + (b div 8 mod 2): positions 2 x and
// 2 x + 1 -- the two halves of a line -- run on XCD x.  (The last, partial group of a grid keeps its order.)
__device__ __forceinline__ unsigned union_xcd_order(const unsigned b, const unsigned nblocks) {
    const unsigned g = b & ~15u;
    if(g + 16u > nblocks) return b;
    const unsigned r = b & 15u;
    return g + ((r & 7u) << 1) + (r >> 3);
}

template <int NC>
struct UnionLds {
    static constexpr int U_WCAP = UnionCfg<NC>::WCAP, U_MAXU = UnionCfg<NC>::MAXU, U_SOLVE = UnionCfg<NC>::SOLVE;
    union {
        float rho[U_WCAP][64];     // scan: rho(cell = lane, candidate slot); +inf = not (or no longer) selected by that cell
        struct {                   // solve (the cells take their rho values into registers first)
            double solve[U_SOLVE]; // column staging of the matrix build, then L_C / 1/diag / L_C^-1 d / B / Schur / d'
            float erho[U_MAXE][64];// rho(cell, extras row)
        } f;
    };
    int wpos[U_WCAP];          // slot -> sorted position of the observation
    int worig[U_WCAP];         // slot -> observation index (tie-break); later: obs - background of the extras (float bits)
#ifdef GPP_UNION_PROFILE
    unsigned prof[24];
#endif
    union {
        struct {
            float4 orec[U_MAXU];   // solve, P build: x, y, z, elevation of the observation of every matrix row
            float olaf[U_MAXU];    //                 its land area fraction
        };
        double late[8][9];         // solve, behind the P build (the records are dead then): (P+R | d) entries of columns 32..39: rows 32..39 and
                                   // the obs - background row  (round 6: overlaid -- 576 B per wave; two waves + the pair block must stay under
                                   // 21 LDS granules of 1 280 B for six workgroups of k_oi_union_pair per CU)
    };
};

// k_oi_union_pair: what the two waves of a workgroup (two neighbouring tiles) tell each other.  See `PAIR` in union_item.
struct PairLds {
    int pos[2][UnionCfg<32>::WCAP];   // wave w, slot s: sorted position of the observation some cell of that tile still holds; -1: slot not live
    unsigned long long core[2];       // slots selected by every updating cell of the wave's own tile
    int state[2];                     // 0: the tile has cells to update and its scan fitted the slots; 1: nothing to update; 2: declined by the scan
    int ok[2];                        // the merged union fits and no cell of this wave has more than U_MAXM extras
    float dmax, dmin;                 // extremes of obs - background over the shared core (oi.cpp:318-334), from the eliminating wave
};

// doubles of the shared-factor area a union of c core and nE extras rows needs (layouts: see the solve).  COLS: the packed-column / packed-row
// layouts (GPP_UNION_COLS, and the LU of the spatially varying form); else the row-packed Cholesky layout of the product.
#ifndef GPP_UNION_COLS
#define GPP_UNION_COLS 0
#endif
template <int NC, bool COLS>
__device__ __forceinline__ int union_solve_doubles(const int c, const int nE) {
    if constexpr(COLS) return c * (c + nE + 1) - c * (c - 1) / 2 + c + nE * nE + nE;   // packed columns (rows of U), 1/diag, Schur complement, d'
    else return c * (c + 1) / 2 + 2 * c + nE * (c | 1) + nE * nE + nE + 3;             // (+ 3: the padding reads of the per-cell finish behind the last row of B)
}
__device__ __forceinline__ double rsqrt_nr(const double a) {
    double rs = __builtin_amdgcn_rsq(a);
    rs = rs * (1.5 - 0.5 * a * rs * rs);
    rs = rs * (1.5 - 0.5 * a * rs * rs);
    return rs;
}

// One wave = one work item.  LIST = false (first pass): the item is a tile.  LIST = true: the items are the 4 children
// of every entry of in_list -- a.level 1: the 16-cell quarters of a declined tile (16 consecutive lanes),
// a.level 2: the 4-cell quarters of a declined 16-cell item.  Smaller items have smaller unions, so almost
// everything ends on this kernel; what it declines at level 2 goes to k_oi (one factorisation per distinct selection).
// List entries: tile (produced by the first pass); tile * 32 + code with code 16..19 = 16-cell item, 0..15 = 4-cell item;
// ~tile = whole tile forwarded unsplit because splitting would not pay (see `forward` below).
// One work item (a tile, or a 16-cell / 4-cell part of one) on one wave.
//
// PAIR (round 6, k_oi_union_pair: first pass, NC = 32): the two waves of a workgroup run two NEIGHBOURING tiles and share ONE factorisation.
// Each wave scans its own tile as before; then both publish which observations their slots hold, both derive the same merged union (core = the
// observations every updating cell of BOTH tiles selected, first; extras behind), and when that fits the limits of the shared factorisation
// (95 % of the headline's pairs: profiles/r05_two_tile_unions.txt) wave 0 loads the records, builds P, eliminates and exports the factor into
// ITS solve area while wave 1 only gathers its cells' G vectors; after a barrier the lanes of both waves finish their cells against that one
// factor.  A pair that does not fit -- or whose partner is absent (odd tile count, a tile the first pass leaves to the list passes), has nothing
// to update, or was declined by its scan -- goes on as two independent single-tile items, exactly the code below without PAIR.
// Barriers (workgroup = the pair): B1 behind the scans (only when both waves run), B2 behind the merge (only when both states are 0), B3 behind
// the export (only when paired); every condition is computed from values both waves read identically, so the counts always agree.
//
// SP (round 6, k_oi_union_sp): a spatially varying Barnes structure -- h, v, w and the localization distance looked up at the FIRST point of
// corr(p1, p2) (src/api/structure.cpp:188-214).  The cells of a tile still select almost the same observations, but P is not symmetric
// (row i carries the scales at observation i, oi.cpp:304-312), so the shared factor is an LU of the core block: rows in lanes as before, no
// pivoting (P + R of a correlation function with mildly varying scales; a vanishing pivot raises the error flag and the call is redone by the
// pivoted LU of k_oi, like a non-positive pivot of the Cholesky form), the rows of U dumped to LDS as they become final -- which is also the
// broadcast of the rank-1 update --, late columns and obs - background substituted against L afterwards, and the per-cell finish sweeps the
// rows of U: increment = (U^-T g) . (L^-1 d).  No variance output (it needs L^-1 g as well: those calls stay on k_oi).
template <bool PLAIN, bool LIST, int NC, bool PAIR = false, bool SP = false>
__device__ __forceinline__ void union_item(const OiArgs& a, int tile, int sub, const int shift, UnionLds<NC>& L, const int lane,
                                           const bool partner = false, const int pw = 0, const int lead = 0, UnionLds<NC>* const Lall = nullptr, PairLds* const PP = nullptr) {
    constexpr int U_WCAP = UnionCfg<NC>::WCAP, U_MAXU = UnionCfg<NC>::MAXU, U_SOLVE = UnionCfg<NC>::SOLVE;
    static_assert(!PAIR || (NC == 32 && !LIST), "the pair form exists for the first pass of the 32-column kernel only");
    static_assert(!SP || (PLAIN && NC == 32 && !PAIR), "the spatially varying form exists for the 32-column Barnes kernel only");
    tile = __builtin_amdgcn_readfirstlane(tile); sub = __builtin_amdgcn_readfirstlane(sub);
#ifdef GPP_UNION_PROFILE
    if(lane < 24) L.prof[lane] = 0u;   // (in LDS: sixteen accumulators in scalar registers spill the kernel into another one)
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
    int cell = -1;
    if(a.tiled2d) {
        int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
        int y = ty * (64 >> a.wshift) + (lane >> a.wshift), x = (tx << a.wshift) + (lane & ((1 << a.wshift) - 1));
        if(y < a.ny && x < a.nx) cell = y * a.nx + x;
    }
    else {
        int c = tile * 64 + lane;
        if(c < a.C) cell = c;
    }
    if(LIST && (lane >> shift) != sub) cell = -1;
    float gx = 0, gy = 0, gz = 0, ge = NAN, gl = NAN, bg = NAN, bvar = 1.0f;
    if(cell >= 0) {
        gx = a.gx[cell]; gy = a.gy[cell]; gz = a.gz[cell]; ge = a.gelev[cell]; gl = a.glaf[cell];
        bg = a.bg[cell];
        if(a.bvar) bvar = a.bvar[cell];
    }
    const bool active = cell >= 0 && d_valid(bg);   // oi.cpp:223
    // (a copy whose scales the optimiser cannot see through when the item runs inside the tile loop of the persistent first pass: their
    //  reciprocals, squares and pruning constants are otherwise invariants of that loop, kept in a dozen vector registers)
    DevStructure st = a.s.st;
    const ScanArgs& sa = a.s;
    float sa_inv_s = sa.inv_s;
    if constexpr(SP) { if(cell >= 0) d_structure_at(st, st.cell_idx ? st.cell_idx[cell] : cell); }   // this cell's scales: rho(cell, observation) takes them at the cell
    else if constexpr(!LIST) asm volatile("" : "+s"(st.h), "+s"(st.v), "+s"(st.w), "+s"(st.R), "+s"(sa_inv_s));

    // ================= candidate scan (the walk of scan_tile; selections kept as rho[slot][lane]) =================
    UPROF(0);   // cell loads issued
    const float R = st.R;
    const int K = sa.K;
    const float h2 = st.h * st.h;
    const bool prune = PLAIN || st.kh == SK_BARNES;
    const float pa = sa.axis_a == 0 ? gx : (sa.axis_a == 1 ? gy : gz);
    const float pb = sa.axis_b == 1 ? gy : (sa.axis_b == 2 ? gz : gx);
    const float amin_t = wave_min(active ? pa : INFINITY), amax_t = wave_max(active ? pa : -INFINITY);
    const float bmin_t = wave_min(active ? pb : INFINITY), bmax_t = wave_max(active ? pb : -INFINITY);
    // Elevation / land-fraction extent of the tile (PLAIN: Barnes factors): a candidate outside it by dz can reach no cell with more than
    // exp(-dz^2 / 2 v^2) of its horizontal factor -- the wave-level prune below adds that to the horizontal distance (round 4: on smooth
    // terrain the worst kept rho of a cell is small, so the horizontal cut alone let 5 x as many candidates through as on flat ground, each
    // evaluated for all 64 cells).  A cell without elevation takes the factor 1: then there is no extent.
    float emin_t = -INFINITY, emax_t = INFINITY, lmin_t = -INFINITY, lmax_t = INFINITY, kv2 = 0.0f, kw2 = 0.0f;
    if constexpr(PLAIN && !SP) {   // (SP: the ratios differ from cell to cell; the horizontal prune alone)
        if(d_valid(st.v) && st.v != 0.0f && d_valid(st.h) && st.h != 0.0f) {
            emin_t = wave_min(active ? (d_valid(ge) ? ge : -INFINITY) : INFINITY); emax_t = wave_max(active ? (d_valid(ge) ? ge : INFINITY) : -INFINITY);
            kv2 = (st.h / st.v) * (st.h / st.v) * 0.9999f;
        }
        if(d_valid(st.w) && st.w != 0.0f && d_valid(st.h) && st.h != 0.0f) {
            lmin_t = wave_min(active ? (d_valid(gl) ? gl : -INFINITY) : INFINITY); lmax_t = wave_max(active ? (d_valid(gl) ? gl : INFINITY) : -INFINITY);
            kw2 = (st.h / st.w) * (st.h / st.w) * 0.9999f;
        }
    }
    int cnt = 0;
    unsigned long long alloc = 0ull;   // allocated slots (wave-uniform)
    bool fb = false;                   // wave-uniform: this tile goes to k_oi
    constexpr unsigned long long FULL = U_WCAP >= 64 ? ~0ull : ((1ull << (U_WCAP & 63)) - 1ull);
    // an empty (never used / evicted / not wanted by this cell) entry is +inf; all slot loops are static so that the
    // slot LDS reads issue back to back (ds_read2st64) instead of one latency per slot
#pragma unroll
    for(int w = 0; w < U_WCAP; ++w) L.rho[w][lane] = INFINITY;
    auto live_mask = [&]() {   // slots some cell still holds
        unsigned long long live = 0ull;
        float rv[U_WCAP];
#pragma unroll
        for(int w = 0; w < U_WCAP; ++w) rv[w] = L.rho[w][lane];
#pragma unroll
        for(int w = 0; w < U_WCAP; ++w) live |= ((wave_ballot(rv[w] < INFINITY) != 0ull) ? 1ull : 0ull) << w;
        return live;
    };
    // The worst kept entry of this lane: smallest rho r0, a slot s0 that holds it, and whether more than one slot does.  Two levels --
    // minima of groups of eight slots, then the eight values of the group that holds the minimum, read again from LDS (its index is
    // the lane's own): ~70 instructions.  One flat pass (compare, select, count for each of the 40 slots, each triple with a wait
    // state behind the compare) was 180, and a tile runs this 5.5 times (once behind the bulk disc, once per eviction).
    auto worst_slot = [&](float& r0, int& s0, bool& multi) {
        constexpr int NG = U_WCAP / 8;
        float gm[NG];
#pragma unroll
        for(int g = 0; g < NG; ++g) {
            float v[8];
#pragma unroll
            for(int j = 0; j < 8; ++j) v[j] = L.rho[8 * g + j][lane];
            gm[g] = fminf(fminf(fminf(v[0], v[1]), fminf(v[2], v[3])), fminf(fminf(v[4], v[5]), fminf(v[6], v[7])));
        }
        r0 = gm[0];
#pragma unroll
        for(int g = 1; g < NG; ++g) r0 = fminf(r0, gm[g]);
        int gs = 0, ng = 0;
#pragma unroll
        for(int g = 0; g < NG; ++g) { const bool eq = gm[g] == r0; gs = eq ? g : gs; ng += eq ? 1 : 0; }
        const float* const grp = &L.rho[0][lane] + gs * (8 * 64);
        float v[8];
#pragma unroll
        for(int j = 0; j < 8; ++j) v[j] = grp[j * 64];
        int sj = 0, nin = 0;
#pragma unroll
        for(int j = 0; j < 8; ++j) { const bool eq = v[j] == r0; sj = eq ? j : sj; nin += eq ? 1 : 0; }
        s0 = 8 * gs + sj;
        multi = ng > 1 || nin > 1;
    };
    if(wave_ballot(active) != 0ull) {
        const float sbin = 1.0f / sa_inv_s;
        int tby0 = (int)floorf((bmin_t - sa.bmin) * sa_inv_s), tby1 = (int)floorf((bmax_t - sa.bmin) * sa_inv_s);
        tby0 = __builtin_amdgcn_readfirstlane(min(max(tby0, 0), sa.nby - 1));
        tby1 = __builtin_amdgcn_readfirstlane(min(max(tby1, tby0), sa.nby - 1));
        int tbx0 = (int)floorf((amin_t - sa.amin) * sa_inv_s), tbx1 = (int)floorf((amax_t - sa.amin) * sa_inv_s);
        tbx0 = __builtin_amdgcn_readfirstlane(min(max(tbx0, 0), sa.nbx - 1));
        tbx1 = __builtin_amdgcn_readfirstlane(min(max(tbx1, tbx0), sa.nbx - 1));
        const float lox = gx - R, hix = gx + R, loy = gy - R, hiy = gy + R, loz = gz - R, hiz = gz + R;
        const float near_r = R - 4.76837158e-7f * (fabsf(gx) + fabsf(gy) + fabsf(gz) + R);   // (see eval)
        float wr = 0.0f;       // worst kept rho, its slot and observation index
        int ws = 0;
        unsigned wo = 0u;
        const float thr2_R = R * R * 1.000001f + 1e-30f;
        float thr2 = active ? thr2_R : -1.0f;
        float bud2 = active ? INFINITY : -1.0f;   // the same threshold without the clamp to R^2: the budget for horizontal + vertical + laf terms (h^2 units)

        // projected (bin-axis) position of the tile centre and its half diagonal: a candidate whose projected distance to
        // the centre exceeds sqrt(largest threshold in the wave) + rad cannot be wanted by any cell (wave-level prune)
        const float ca = 0.5f * (amin_t + amax_t), cb = 0.5f * (bmin_t + bmax_t);
        const float rad = 0.5f * d_sqrt_raw((amax_t - amin_t) * (amax_t - amin_t) + (bmax_t - bmin_t) * (bmax_t - bmin_t)) * 1.001f + 1e-30f;
        auto proj_d2 = [&](const float4& rec) {
            const float ra = sa.axis_a == 0 ? rec.x : (sa.axis_a == 1 ? rec.y : rec.z);
            const float rb = sa.axis_b == 1 ? rec.y : (sa.axis_b == 2 ? rec.z : rec.x);
            return (ra - ca) * (ra - ca) + (rb - cb) * (rb - cb);
        };
        auto wave_lim2 = [&]() {   // < 0: nothing can be wanted any more
            const float t2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wave_max(thr2))));
            if(t2 < 0.0f) return -1.0f;
            const float lim = (d_sqrt_raw(t2) + rad) * 1.0001f;
            return lim * lim;
        };

        // wave-level prune of the candidate a lane holds (pd: its projected distance^2 to the tile centre): inside the horizontal limit, and --
        // once every cell of the wave has its max_points -- inside the largest budget with the elevation / laf distance to the tile's extent added
        const bool eprune = PLAIN && (kv2 != 0.0f || kw2 != 0.0f);   // (wave-uniform; flat ground pays nothing for the extent)
        auto wave_bud2 = [&]() { return eprune ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wave_max(bud2)))) : INFINITY; };
        auto keep = [&](const float pd, const float4& rec, const float2& met, const float lim2, const float B2) {
            bool k = pd <= lim2;
            if(eprune && B2 < INFINITY) {
                const float sd = fmaxf(d_sqrt_raw(pd) * 0.9999f - rad, 0.0f);
                const float dz = fmaxf(fmaxf(emin_t - rec.w, rec.w - emax_t), 0.0f), dl = fmaxf(fmaxf(lmin_t - met.x, met.x - lmax_t), 0.0f);   // (NaN: no distance)
                k = k && (sd * sd + kv2 * dz * dz + kw2 * dl * dl <= B2);
            }
            return k;
        };

        // the candidates `mask` of one chunk of 64 records (lane c holds record c: rec, met, sorted position posv)
        // rho(cell, candidate c of the chunk) as corr_background gives it, 0 when the candidate is not usable for this cell.
        // PLAIN: straight-line code (selects instead of branches) so that two candidates evaluated back to back interleave.
        const bool hv = d_valid(st.v) && st.v != 0.0f, hw = d_valid(st.w) && st.w != 0.0f, hh = d_valid(st.h) && st.h != 0.0f;
        const double rh = hh ? 1.0 / (double)st.h : 0.0, rv_ = hv ? 1.0 / (double)st.v : 0.0, rw_ = hw ? 1.0 / (double)st.w : 0.0;
        // (SP: the scales are this lane's; the tests around the vertical / laf factors stay wave-uniform -- "some cell has one")
        const bool hvu = SP ? wave_ballot(hv) != 0ull : hv, hwu = SP ? wave_ballot(hw) != 0ull : hw;
        auto eval = [&](const float4& rec, const float2& met, const int c) {
            const float ox = readlane_f(rec.x, c), oy = readlane_f(rec.y, c), oz = readlane_f(rec.z, c);
            const float dx = ox - gx, dy = oy - gy, dz = oz - gz;
            float d2 = dx * dx + dy * dy;
            d2 = d2 + dz * dz;
            const float dist = d_sqrt_cr(d2);
            bool ok = d2 <= thr2 && dist <= R;   // within_radius (kdtree.cpp:255), the cut inside corr (structure.cpp:216)
            // The strictly-inside box of the radius query (kdtree.cpp:46,53; its corners gx -+ R are float32 roundings) is implied by
            // dist <= R unless the candidate sits within a few ulps OF THE COORDINATES of the radius: dist <= near_r = R - 2^-21 (|gx| + |gy| +
            // |gz| + R) gives |ox - gx| <= dist (1 + 2^-22) < R - 2^-24 (|gx| + R) <= the distance of either rounded corner, on every axis.
            // The six compares are made only when some cell of the wave holds a candidate that close to R (pruning by the worst kept rho
            // keeps the candidates far inside: never on the headline workload).
            if(wave_ballot(ok && dist > near_r) != 0ull) ok = ok && ox > lox && ox < hix && oy > loy && oy < hiy && oz > loz && oz < hiz;
            float rho = 0.0f;
            if constexpr(PLAIN) {
                rho = hh ? d_barnes_rho_flat(dist, rh) : 1.0f;
                if(hvu) {
                    const float oe = readlane_f(rec.w, c);
                    if(d_valid(oe)) { const float f = d_barnes_rho_flat(ge - oe, rv_); rho = (hv && d_valid(ge)) ? rho * f : rho; }
                }
                if(hwu) {
                    const float ol = readlane_f(met.x, c);
                    if(d_valid(ol)) { const float f = d_barnes_rho_flat(gl - ol, rw_); rho = (hw && d_valid(gl)) ? rho * f : rho; }
                }
                rho = ok ? rho : 0.0f;
            }
            else if(ok) {
                const float oe = readlane_f(rec.w, c), ol = readlane_f(met.x, c);
                rho = (st.cv && dist <= st.cv_dist) ? 0.0f : d_rho(st.kh, dist, st.h);   // corr_background
                if(d_valid(ge) && d_valid(oe)) rho *= d_rho(st.kv, ge - oe, st.v);
                if(d_valid(gl) && d_valid(ol)) rho *= d_rho(st.kw, gl - ol, st.w);
            }
            return rho;
        };

        // PLAIN: EV candidates at once.  One candidate is a chain of ~35 dependent operations (chord, correctly rounded root and quotient, the
        // table exp with its LDS lookup): evaluated one after the other a wave spends the scan waiting for its own results -- the per-phase
        // clocks did not change when the other waves of the SIMD idled (round 6, the pair kernel's waiting wave).  Independent chains side by
        // side fill those slots.  Same values as eval().
#ifndef GPP_UNION_EV
#define GPP_UNION_EV 2    // (measured: 4 side by side changes nothing -- 28.6 k clocks for the ~30 bulk candidates of a tile either way, profiles/r06_union_phases.txt:
                          //  with three waves per SIMD the scan is bound by instruction issue, not by the chains -- and costs 24 registers)
#endif
        constexpr int EV = GPP_UNION_EV;
        auto eval_n = [&](const float4& rec, const float2& met, const int (&cc)[EV], float (&rho)[EV]) {
            float dist[EV];
            bool ok[EV];
            bool anynear = false;
#pragma unroll
            for(int q = 0; q < EV; ++q) {
                const float ox = readlane_f(rec.x, cc[q]), oy = readlane_f(rec.y, cc[q]), oz = readlane_f(rec.z, cc[q]);
                const float dx = ox - gx, dy = oy - gy, dz = oz - gz;
                float d2 = dx * dx + dy * dy;
                d2 = d2 + dz * dz;
                dist[q] = d_sqrt_cr(d2);
                ok[q] = d2 <= thr2 && dist[q] <= R;
                anynear = anynear || (ok[q] && dist[q] > near_r);
            }
            if(wave_ballot(anynear) != 0ull) {   // (see eval: the strictly-inside box of the radius query, only for candidates within ulps of R)
#pragma unroll
                for(int q = 0; q < EV; ++q) {
                    const float ox = readlane_f(rec.x, cc[q]), oy = readlane_f(rec.y, cc[q]), oz = readlane_f(rec.z, cc[q]);
                    ok[q] = ok[q] && ox > lox && ox < hix && oy > loy && oy < hiy && oz > loz && oz < hiz;
                }
            }
            // (the wave-uniform tests around whole groups of EV chains, not inside them: a branch ends the block the scheduler interleaves in)
#pragma unroll
            for(int q = 0; q < EV; ++q) rho[q] = 1.0f;
            if(SP ? wave_ballot(hh) != 0ull : hh) {
#pragma unroll
                for(int q = 0; q < EV; ++q) { const float f = d_barnes_rho_flat(dist[q], rh); rho[q] = hh ? f : 1.0f; }
            }
            if(hvu) {
#pragma unroll
                for(int q = 0; q < EV; ++q) {
                    const float oe = readlane_f(rec.w, cc[q]);
                    const float f = d_barnes_rho_flat(d_valid(oe) ? ge - oe : 0.0f, rv_);
                    rho[q] = (hv && d_valid(oe) && d_valid(ge)) ? rho[q] * f : rho[q];
                }
            }
            if(hwu) {
#pragma unroll
                for(int q = 0; q < EV; ++q) {
                    const float ol = readlane_f(met.x, cc[q]);
                    const float f = d_barnes_rho_flat(d_valid(ol) ? gl - ol : 0.0f, rw_);
                    rho[q] = (hw && d_valid(ol) && d_valid(gl)) ? rho[q] * f : rho[q];
                }
            }
#pragma unroll
            for(int q = 0; q < EV; ++q) rho[q] = ok[q] ? rho[q] : 0.0f;
        };

        // the candidates `mask` of one chunk of 64 records (lane c holds record c: rec, met, sorted position posv)
        auto run_chunk = [&](const float4 rec, const float2 met, const int posv, unsigned long long mask) {
            while(mask != 0ull && !fb) {
                const int c = __builtin_ctzll(mask);
                mask &= mask - 1ull;
                const float rho = eval(rec, met, c);
                if(UNION_STATS && lane == 0) atomicAdd(&a.counters[11], 1ull);   // candidates evaluated behind the wave-level prune
                {
                const unsigned orig = (unsigned)__builtin_amdgcn_readlane(__float_as_int(met.y), c);
                // oi.cpp:253 (rho > 0) and :262-273 (keep the max_points largest, ties -> lower observation index)
                const bool want = rho > 0.0f && (cnt < K || rho > wr || (rho == wr && orig < wo));
                if(wave_ballot(want) != 0ull) {
                    if(alloc == FULL) alloc = live_mask();   // free the slots no cell holds any more
                    if(UNION_STATS && lane == 0) atomicAdd(&a.counters[9], 1ull);
                    if(UNION_STATS && wave_ballot(want && cnt >= K) != 0ull && lane == 0) atomicAdd(&a.counters[10], 1ull);
                    if(alloc == FULL) { fb = true; if(UNION_STATS && lane == 0) atomicAdd(&a.counters[4], 1ull); return; }
                    const int slot = __builtin_ctzll(~alloc);
                    alloc |= 1ull << slot;
                    if(lane == 0) { L.wpos[slot] = __builtin_amdgcn_readlane(posv, c); L.worig[slot] = (int)orig; }
                    L.rho[slot][lane] = want ? rho : INFINITY;
                    if(want) {
                        if(cnt < K) {
                            if(cnt == 0 || rho < wr || (rho == wr && orig > wo)) { wr = rho; ws = slot; wo = orig; }
                            cnt++;
                        }
                        else {
                            L.rho[ws][lane] = INFINITY;
                            float r0;
                            int s0;
                            bool multi;
                            worst_slot(r0, s0, multi);
                            wr = r0; ws = s0;
                            if(multi) {   // equal rho: the higher observation index is the worse one
                                unsigned bo = 0u;
                                for(unsigned long long mm = alloc; mm; mm &= mm - 1ull) {
                                    const int w = __builtin_ctzll(mm);
                                    if(L.rho[w][lane] == r0) {
                                        const unsigned o = (unsigned)L.worig[w];
                                        if(o >= bo) { bo = o; ws = w; }
                                    }
                                }
                                wo = bo;
                            }
                            else wo = (unsigned)L.worig[ws];
                        }
                        if(prune && cnt == K) { bud2 = -2.0f * h2 * logf(wr) * 1.00002f + 2e-5f * h2; thr2 = fminf(thr2_R, bud2); }
                    }
                }
            }
                }
        };

        UPROF(1);   // bbox reductions, LDS init
        // ---- phase 1: the square of bins around the tile, all records loaded at once (up to 3 chunks of 64), visited in
        //      rings of growing projected distance from the tile centre so that the thresholds tighten as early as possible.
        //      The order only changes the amount of work, never the selection.
        int q = sa.q0;
        int sx0 = 0, sx1 = -1, sy0 = 1, sy1 = 0;   // the square phase 1 covered (empty: sy0 > sy1)
        {
            int js = 0, len = 0, pre = 0, total = 0, nrows = 0;
            bool have = false;
            for(;; --q) {
                const int x0 = max(tbx0 - q, 0), x1 = min(tbx1 + q, sa.nbx - 1);
                const int y0 = max(tby0 - q, 0), y1 = min(tby1 + q, sa.nby - 1);
                nrows = y1 - y0 + 1;
                if(nrows <= 64) {
                    js = 0; len = 0;
                    if(lane < nrows) {
                        const int rb = (y0 + lane) * sa.nbx;
                        js = sa.bin_start[rb + x0];
                        len = sa.bin_start[rb + x1 + 1] - js;
                    }
                    pre = wave_scan_add(len);
                    total = __builtin_amdgcn_readlane(pre, 63);
                    pre -= len;
                    if(total <= 192) { have = true; sx0 = x0; sx1 = x1; sy0 = y0; sy1 = y1; break; }
                }
                if(q == 0) break;
            }
            if(have && total > 0) {
                const int nchunk = (total + 63) >> 6;
                int row0 = 0, row1 = 0, row2 = 0;
                for(int r = 1; r < nrows; ++r) {
                    const int pr = __builtin_amdgcn_readlane(pre, r);
                    if(lane >= pr) row0 = r;
                    if(lane + 64 >= pr) row1 = r;
                    if(lane + 128 >= pr) row2 = r;
                }
                float4 rec0 = make_float4(NAN, 0, 0, NAN), rec1 = rec0, rec2 = rec0;
                float2 met0 = make_float2(NAN, 0), met1 = met0, met2 = met0;
                int pos0 = 0, pos1 = 0, pos2 = 0;
                float pd0 = INFINITY, pd1 = INFINITY, pd2 = INFINITY;
                {
                    pos0 = __shfl(js, row0) + lane - __shfl(pre, row0);
                    if(lane < total) { rec0 = sa.pgeo[pos0]; met0 = sa.smeta[pos0]; }
                    if(nchunk > 1) {
                        pos1 = __shfl(js, row1) + lane + 64 - __shfl(pre, row1);
                        if(lane + 64 < total) { rec1 = sa.pgeo[pos1]; met1 = sa.smeta[pos1]; }
                    }
                    if(nchunk > 2) {
                        pos2 = __shfl(js, row2) + lane + 128 - __shfl(pre, row2);
                        if(lane + 128 < total) { rec2 = sa.pgeo[pos2]; met2 = sa.smeta[pos2]; }
                    }
                    if(lane < total) pd0 = proj_d2(rec0);
                    if(lane + 64 < total) pd1 = proj_d2(rec1);
                    if(lane + 128 < total) pd2 = proj_d2(rec2);
                }
                UPROF(2);   // bin_start, prefix, chunk loads
                // ring k: projected distance in [ring_r0 + (k-1) ring_dr, ring_r0 + k ring_dr), the last ring takes the rest
                constexpr int NR = 4;
                float lo2 = -1.0f;
                // Bulk mode: while the rings visited so far hold at most max_points candidates no cell can have to choose, so
                // a candidate simply takes the next slot (rho, or +inf where it is unusable for that cell): no selection logic.
                int nb = 0;
                bool bulk = true;
                auto end_bulk = [&]() {   // worst kept entry of every cell after the bulk rings
                    bulk = false;
                    alloc = nb >= 64 ? ~0ull : ((1ull << nb) - 1ull);
                    float r0;
                    int s0;
                    bool multi;
                    worst_slot(r0, s0, multi);
                    wr = r0; ws = s0;
                    wo = (unsigned)L.worig[s0];
                    if(cnt > 0 && multi) {   // equal rho: the higher observation index is the worse one
                        unsigned bo = 0u;
                        for(unsigned long long mm = alloc; mm; mm &= mm - 1ull) {
                            const int w = __builtin_ctzll(mm);
                            if(L.rho[w][lane] == r0) {
                                const unsigned o = (unsigned)L.worig[w];
                                if(o >= bo) { bo = o; ws = w; }
                            }
                        }
                        wo = bo;
                    }
                    if(prune && cnt == K) { bud2 = -2.0f * h2 * logf(wr) * 1.00002f + 2e-5f * h2; thr2 = fminf(thr2_R, bud2); }
                };
                // radius^2 (projected) holding at most min(max_points, slots) records: bisection with wave-wide counts
                const int kb = min(K, U_WCAP);
                float tlo = 0.0f;
                {
                    float thi = sa.ring_r0 * sa.ring_r0;   // 1.5 x the expected distance of the max_points-th nearest observation
                    const float lim2 = wave_lim2();
                    thi = fminf(thi, lim2);
                    auto count_le = [&](const float t) { return __popcll(wave_ballot(pd0 <= t)) + __popcll(wave_ballot(pd1 <= t)) + __popcll(wave_ballot(pd2 <= t)); };
                    if(count_le(thi) <= kb) tlo = thi;
                    else {
                        for(int it = 0; it < 9; ++it) {
                            const float mid = 0.5f * (tlo + thi);
                            if(count_le(mid) <= kb) tlo = mid; else thi = mid;
                        }
                    }
                }
                UPROF(16);  // bisection
                for(int ring = -1; ring < NR && !fb; ++ring) {   // ring -1: the bulk disc
                    const float hi = d_sqrt_raw(tlo) + (float)(ring + 1) * sa.ring_dr;
                    const float hi2 = ring < 0 ? tlo : ((ring == NR - 1) ? INFINITY : hi * hi);
                    const float lim2 = wave_lim2();
                    if(lim2 < 0.0f || lo2 >= lim2) break;
                    const float B2 = wave_bud2();
                    const unsigned long long m0 = wave_ballot(pd0 > lo2 && pd0 <= hi2 && keep(pd0, rec0, met0, lim2, B2));
                    const unsigned long long m1 = wave_ballot(pd1 > lo2 && pd1 <= hi2 && keep(pd1, rec1, met1, lim2, B2));
                    const unsigned long long m2 = wave_ballot(pd2 > lo2 && pd2 <= hi2 && keep(pd2, rec2, met2, lim2, B2));
                    if(ring >= 0 && bulk) { UPROF(17); end_bulk(); UPROF(18); }   // 17: bulk evaluations, 18: end_bulk
                    for(int k = 0; k < nchunk && !fb; ++k) {
                        const float4 rec = k == 0 ? rec0 : (k == 1 ? rec1 : rec2);
                        const float2 met = k == 0 ? met0 : (k == 1 ? met1 : met2);
                        const int posv = k == 0 ? pos0 : (k == 1 ? pos1 : pos2);
                        const unsigned long long mk = k == 0 ? m0 : (k == 1 ? m1 : m2);
                        if(GPP_DBG(a, 8)) continue;
                        if(bulk) {
                            if((mk >> lane) & 1ull) {   // the holder of a candidate records where it lives
                                const int slot = nb + __popcll(mk & ((1ull << lane) - 1ull));
                                L.wpos[slot] = posv;
                                L.worig[slot] = __float_as_int(met.y);
                            }
                            if constexpr(PLAIN && EV > 1) {
                                for(unsigned long long mm = mk; mm != 0ull;) {
                                    int cc[EV], n = 0;
#pragma unroll
                                    for(int q = 0; q < EV; ++q) {
                                        cc[q] = mm != 0ull ? __builtin_ctzll(mm) : cc[0];   // (a short last group evaluates its first candidate again)
                                        n += mm != 0ull ? 1 : 0;
                                        mm &= mm - 1ull;
                                    }
                                    float rho[EV];
                                    eval_n(rec, met, cc, rho);
#pragma unroll
                                    for(int q = 0; q < EV; ++q) {
                                        if(q < n) {
                                            const bool ok = rho[q] > 0.0f;   // oi.cpp:253
                                            L.rho[nb][lane] = ok ? rho[q] : INFINITY;
                                            cnt += ok ? 1 : 0;
                                            nb++;
                                        }
                                    }
                                }
                            }
                            else
                            for(unsigned long long mm = mk; mm != 0ull; mm &= mm - 1ull) {
                                const int c = __builtin_ctzll(mm);
                                const float rho = eval(rec, met, c);
                                const bool ok = rho > 0.0f;   // oi.cpp:253
                                L.rho[nb][lane] = ok ? rho : INFINITY;
                                cnt += ok ? 1 : 0;
                                nb++;
                            }
                        }
                        else run_chunk(rec, met, posv, mk);
                    }
                    lo2 = hi2;
                }
                if(bulk) { UPROF(17); end_bulk(); UPROF(18); }
            }
        }

        UPROF(3);   // ring loop
        // ---- phase 2: every remaining bin that can still matter, rows centre-out (x-extent and stop test from the largest
        //      threshold in the wave), skipping the square phase 1 covered.  Round 4: 32 bin rows at a time, one lane per row segment (left /
        //      right of the phase-1 square: lanes 0..31 / 32..63) -- the bin offsets of all of them in one round of loads, their records packed
        //      into chunks of 64.  Row by row, every band cost two dependent trips to memory for a handful of records (smooth terrain: the
        //      thresholds stay wide, dozens of bands, 34 % of the kernel for 16 records per tile).  The order only changes the amount of work.
        for(int b = 0; !fb && !GPP_DBG(a, 4); ++b) {
            const float t2 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wave_max(thr2))));
            if(t2 < 0.0f) break;
            const int n0 = tby1 - tby0 + 1;
            const int idx = 32 * b + (lane & 31), seg = lane >> 5;
            const int jr = idx - n0;
            const int r = idx < n0 ? 0 : 1 + (jr >> 1);
            const int row = idx < n0 ? tby0 + idx : ((jr & 1) ? tby1 + r : tby0 - r);
            const float gap = (r > 1) ? (float)(r - 1) * sbin * 0.999f : 0.0f;   // min projected distance tile -> row band r
            bool v = gap * gap <= t2 && row >= 0 && row < sa.nby;
            // bins are assigned with the same monotone float expression, so no extra bin is needed once wx is padded
            const float wx = d_sqrt_raw(fmaxf(t2 - gap * gap, 0.0f)) * 1.0001f + 1e-3f * sbin;
            int x0 = (int)floorf((amin_t - wx - sa.amin) * sa_inv_s), x1 = (int)floorf((amax_t + wx - sa.amin) * sa_inv_s);
            x0 = min(max(x0, 0), sa.nbx - 1);
            x1 = min(max(x1, x0), sa.nbx - 1);
            int xa = x0, xb = x1;
            if(row >= sy0 && row <= sy1) {
                if(seg == 0) xb = min(x1, sx0 - 1);
                else xa = max(x0, sx1 + 1);
            }
            else if(seg == 1) v = false;
            v = v && xa <= xb;
            int js = 0, len = 0;
            if(v) { js = sa.bin_start[row * sa.nbx + xa]; len = sa.bin_start[row * sa.nbx + xb + 1] - js; }
            int pre = wave_scan_add(len);
            const int total = __builtin_amdgcn_readlane(pre, 63);
            pre -= len;
            for(int base = 0; base < total && !fb; base += 64) {
                const int g = base + lane;
                int pos = -1;
                for(unsigned long long mm = wave_ballot(len > 0); mm != 0ull; mm &= mm - 1ull) {   // the segment of this lane's record
                    const int sl = __builtin_ctzll(mm);
                    const int p = __builtin_amdgcn_readlane(pre, sl), l = __builtin_amdgcn_readlane(len, sl);
                    if(p + l <= base) continue;
                    if(p >= base + 64) break;
                    const int j = __builtin_amdgcn_readlane(js, sl);
                    if(g >= p && g < p + l) pos = j + g - p;
                }
                float4 rec = make_float4(NAN, 0, 0, NAN);
                float2 met = make_float2(NAN, 0);
                float pd = INFINITY;
                if(pos >= 0) { rec = sa.pgeo[pos]; met = sa.smeta[pos]; pd = proj_d2(rec); }
                const float lim2 = wave_lim2();
                const float B2 = wave_bud2();
                if(UNION_STATS && lane == 0) atomicAdd(&a.counters[12], (unsigned long long)min(64, total - base));   // records loaded in phase 2
                run_chunk(rec, met, pos, wave_ballot(keep(pd, rec, met, lim2, B2)));
            }
            // the next 32 rows: only if their first band can still matter (rows of the tile itself, band 0, always do)
            const int nidx = 32 * (b + 1);
            if(nidx >= n0) {
                const int rn = 1 + ((nidx - n0) >> 1);
                const float t2n = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wave_max(thr2))));
                const float gapn = (rn > 1) ? (float)(rn - 1) * sbin * 0.999f : 0.0f;
                if(t2n < 0.0f || gapn * gapn > t2n) break;
                if(tby0 - rn < 0 && tby1 + rn >= sa.nby) break;
            }
        }
    }

    UPROF(4);   // phase 2
    // ================= classification: union, core, extras ========================================================
    const unsigned long long upd = wave_ballot(cnt > 0);
    unsigned long long coreM = 0ull, extM = 0ull;
    int c = 0, nE = 0, u = 0;
    int m = 0;             // this cell's extras: count and their indices (4 bits each)
    unsigned elist = 0u;
    if(!fb && upd != 0ull) {
        float rv[U_WCAP];
#pragma unroll
        for(int w = 0; w < U_WCAP; ++w) rv[w] = L.rho[w][lane];
        // (selects, not branches: written with `if` this loop compiled to two scalar branches per slot -- eighty instruction-buffer
        //  refills per tile)
#pragma unroll
        for(int w = 0; w < U_WCAP; ++w) {
            const unsigned long long mk = wave_ballot(rv[w] < INFINITY) & upd;
            const unsigned long long isc = (mk == upd) ? 1ull : 0ull, ise = (mk != 0ull) ? (isc ^ 1ull) : 0ull;
            coreM |= isc << w;
            extM |= ise << w;
        }
    }
    // row i of the shared factorisation = lane i: the slot of THIS wave that holds its observation (-1: no cell of this tile selected it;
    // pairs only) and the observation's sorted position
    int myslot = 0, mypos = 0;
    bool paired = false;
    if constexpr(PAIR) {
        if(partner) {
            PairLds& P = *PP;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if(lane < U_WCAP) P.pos[pw][lane] = (((coreM | extM) >> lane) & 1ull) ? L.wpos[lane] : -1;
            if(lane == 0) { P.core[pw] = coreM; P.state[pw] = fb ? 2 : (upd == 0ull ? 1 : 0); }
            UPROF(11);  // (classification of the own slots, publishing)
            __syncthreads();                                                                      // B1
            UPROF(12);  // B1 wait
            if(P.state[0] == 0 && P.state[1] == 0) {
                // the same merge on both waves.  Lane s looks at slot s of wave 0 (A) and slot s of wave 1 (B).
                const int pa = lane < U_WCAP ? P.pos[0][lane] : -1, pb = lane < U_WCAP ? P.pos[1][lane] : -1;
                int partB = -1, partA = -1;   // the slot of B that holds the observation of A's slot `lane`, and the other way round
#pragma unroll
                for(int t = 0; t < U_WCAP; ++t) {
                    const int qa = P.pos[0][t], qb = P.pos[1][t];
                    partB = (qb == pa) ? t : partB;
                    partA = (qa == pb) ? t : partA;
                }
                partB = pa >= 0 ? partB : -1;
                partA = pb >= 0 ? partA : -1;
                const unsigned long long coreA = P.core[0], coreB = P.core[1];
                const unsigned long long liveA = wave_ballot(pa >= 0);
                // core of the pair: selected by every updating cell of both tiles; extras: the other live slots of A, then what only B holds
                const unsigned long long pcore = wave_ballot(pa >= 0 && ((coreA >> lane) & 1ull) != 0ull && partB >= 0 && ((coreB >> (partB & 63)) & 1ull) != 0ull);
                const unsigned long long pextA = liveA & ~pcore;
                const unsigned long long bonly = wave_ballot(pb >= 0 && partA < 0);
                const int c2 = __popcll(pcore), nEa = __popcll(pextA), nE2 = nEa + __popcll(bonly), u2 = c2 + nE2;
                const bool fits = u2 <= U_MAXU && nE2 <= U_MAXE && union_solve_doubles<NC, (NC == 32 && GPP_UNION_COLS != 0) || SP>(c2, nE2) <= U_SOLVE;
                int sA = -1, sB = -1;
                {
                    const unsigned long long mk = lane < c2 ? pcore : (lane < c2 + nEa ? pextA : bonly);
                    const int idx = lane < c2 ? lane : (lane < c2 + nEa ? lane - c2 : lane - c2 - nEa);
                    const int sl = nth_set_bit(mk, idx);
                    if(lane < c2 + nEa) sA = sl; else if(lane < u2) sB = sl;
                }
                const int pbs = __shfl(partB, sA < 0 ? 0 : sA);
                if(sA >= 0) sB = pbs;
                const int posA = P.pos[0][sA < 0 ? 0 : sA], posB = P.pos[1][sB < 0 ? 0 : sB];
                const int slot2 = pw == 0 ? sA : sB, pos2 = sA >= 0 ? posA : posB;
                // this cell's extras among the rows c2 .. u2 - 1
                int m2 = 0;
                unsigned el2 = 0u;
                if(fits) {
#pragma unroll
                    for(int ai = 0; ai < U_MAXE; ++ai) {
                        if(ai < nE2) {
                            const int sl = __builtin_amdgcn_readlane(slot2, c2 + ai);
                            const bool sel = sl >= 0 && L.rho[sl < 0 ? 0 : sl][lane] < INFINITY;
                            if(sel) { el2 |= (unsigned)ai << (4 * (m2 & 7)); m2++; }
                        }
                    }
                }
                const bool ok = fits && wave_ballot(m2 > U_MAXM) == 0ull;
                if(lane == 0) P.ok[pw] = ok ? 1 : 0;
                UPROF(13);  // merge
                __syncthreads();                                                                  // B2
                UPROF(14);  // B2 wait
                paired = P.ok[0] != 0 && P.ok[1] != 0;
                if(paired) { c = c2; nE = nE2; u = u2; m = m2; elist = el2; myslot = slot2; mypos = pos2; }
            }
        }
    }
    if(!paired && !fb && upd != 0ull) {
        c = __popcll(coreM); nE = __popcll(extM); u = c + nE;
        if(u > U_MAXU || nE > U_MAXE || union_solve_doubles<NC, (NC == 32 && GPP_UNION_COLS != 0) || SP>(c, nE) > U_SOLVE) {
            fb = true;
            if(UNION_STATS && lane == 0) atomicAdd(&a.counters[u > U_MAXU ? 5 : (nE > U_MAXE ? 6 : 7)], 1ull);
        }
        else {
            int ai = 0;
            for(unsigned long long mm = extM; mm; mm &= mm - 1ull, ++ai) {
                const int w = __builtin_ctzll(mm);
                if(L.rho[w][lane] < INFINITY) { elist |= (unsigned)ai << (4 * (m & 7)); m++; }
            }
            if(wave_ballot(m > U_MAXM) != 0ull) { fb = true; if(UNION_STATS && lane == 0) atomicAdd(&a.counters[8], 1ull); }
        }
    }
    if(fb) {
        if(lane == 0) a.out_list[atomicAdd(a.out_count, 1)] = !LIST ? tile : (tile * 32 + (a.level == 1 ? 16 + sub : sub));
        return;
    }
    const bool solver = !PAIR || !paired || pw == lead;         // this wave builds and eliminates (wave-uniform)
    UnionLds<NC>& LS = (PAIR && paired) ? Lall[lead] : L;      // where the shared factor lives
    UPROF(5);   // classification
    float res_out = bg, res_var = bvar;   // oi.cpp:198-199
    if(upd != 0ull && !GPP_DBG(a, 1)) {
        // ============= shared factorisation: rows 0..c-1 core, c..u-1 extras, lane 63 = obs - background ==========
        if(!paired) {
            myslot = lane < c ? nth_set_bit(coreM, lane) : (lane < u ? nth_set_bit(extM, lane - c) : 0);
            mypos = L.wpos[myslot];
        }
        int oorig = 0;     // SP: the observation (original order) of this row -- its scales are looked up there
        if constexpr(SP) oorig = L.worig[myslot];
        float4 o0 = make_float4(0, 0, 0, NAN), o1 = make_float4(NAN, 0, 0, 0);
        if(solver && lane < u) {
            o0 = sa.pgeo[mypos];
            o1 = a.saux[mypos];
        }
        const float dpf = (float)((double)o1.y - (double)o1.z);   // obs - background at the observation (oi.cpp:293)
        // this cell's rho for every row of the union (lG, oi.cpp:296); afterwards the rho slots are dead and the
        // shared-factor area takes their place
        float gf[NC];
        {
            constexpr int NLATE = U_MAXU > NC ? U_MAXU - NC : 1;
            float gx8[NLATE];   // rows NC.. are always extras (c <= NC): they only pass through
            // (PAIR: a row no cell of this tile selected has no slot here; it is an extras row none of these cells lists)
            auto rho_of_row = [&](const int k) {
                const int sl = __builtin_amdgcn_readlane(myslot, k);
                if constexpr(PAIR) return sl >= 0 ? L.rho[sl][lane] : 0.0f;
                else return L.rho[sl][lane];
            };
#pragma unroll
            for(int k = 0; k < NC; ++k) gf[k] = (k < u) ? rho_of_row(k) : 0.0f;
#pragma unroll
            for(int k = NC; k < U_MAXU; ++k) gx8[k - NC] = (k < u) ? rho_of_row(k) : 0.0f;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for(int k = 0; k < NC; ++k) if(k >= c && k < u) L.f.erho[k - c][lane] = gf[k];
#pragma unroll
            for(int k = NC; k < U_MAXU; ++k) if(k < u) L.f.erho[k - c][lane] = gx8[k - NC];
        }
        if(solver && lane >= c && lane < u) L.worig[lane - c] = __float_as_int(dpf);
        UPROF(6);   // observation records of the union
        // layout of the shared-factor area (doubles).
        // NC = 64 (rows packed): L_C rows, 1/diag, L_C^-1 d, B rows (stride bs), Schur complement, d'.
        // NC = 32 (round 6, COLUMNS packed): column j < c of the factor of [core | extras | obs - background] as the elimination leaves it --
        //   rows j .. u-1, then the entry of the obs - background row -- at CO(j) = j (u + 1) - j (j - 1) / 2; behind the columns 1/diag, the Schur
        //   complement of the extras, d'.  The elimination writes every column ONCE, to its place (it has to go through LDS anyway, for the
        //   broadcast of the rank-1 update), and the per-cell substitution walks the columns: no export pass, no dependent dot product per row.
        // Built and measured in round 6 (GPP_UNION_COLS=1, with the two-column elimination steps and the column sweep of the per-cell finish below):
        // 5.05 ms per headline step against 4.42 with the row form -- elimination 28.6 k instead of 21.9 k clocks per tile, per-cell finish 21.7 k
        // instead of 14.3 k (profiles/r06_union_phases.txt).  Half as many serial steps bought nothing: the kernel is bound by the instructions it
        // issues (VALU 78 % busy), and the column forms issue more of them (496 instead of 378 multiply-adds in the sweep, a per-lane LDS read per
        // extra and column).  The row form stays the product.
        constexpr bool COLS = NC == 32 && GPP_UNION_COLS != 0;
        const int FAC = c * (u + 1) - c * (c - 1) / 2;
        const int oL = 0, oI = (COLS || SP) ? FAC : c * (c + 1) / 2, oZ = oI + c, oB = oZ + c, bs = c | 1, oS = (COLS || SP) ? oI + c : oB + nE * bs, oD = oS + nE * nE;
        double* const sv = LS.f.solve;
        bool bad = false;
        float maxInc = -INFINITY, minInc = INFINITY;
        double inc = 0.0, a00 = 0.0;
        const int mmax = __builtin_amdgcn_readfirstlane((int)wave_max((float)m));
        if constexpr(SP) {
            // ================= spatially varying structure: LU of the core block, rows of U in LDS (layout: row j < c of U -- columns j .. u-1, then
            // (L^-1 d)_j -- at CO(j) = j (u + 1) - j (j - 1) / 2; behind the rows 1 / u_jj, the Schur complement of the extras (nE x nE, both
            // triangles), d') ====================================================================================================================
            const DevStructure& su = a.s.st;
            float* const colbuf = reinterpret_cast<float*>(L.f.solve);     // [u][U_MAXU] floats: P(row i, column p) at [p][i]  (6 400 B)
            double* const rsc = L.f.solve + 800;                           // reciprocal scales of every row: 1/h [40], 1/v [40], 1/w [40] (0: that factor is 1)
            float* const rR = reinterpret_cast<float*>(L.f.solve + 920);   // localization distance of every row [40]
            if(lane < u) {
                const int fi = su.obs_idx[oorig];
                const float h_i = su.fh[fi], v_i = su.fv[fi], w_i = su.fw[fi];
                rsc[lane] = (d_valid(h_i) && h_i != 0.0f) ? 1.0 / (double)h_i : 0.0;
                rsc[40 + lane] = (d_valid(v_i) && v_i != 0.0f) ? 1.0 / (double)v_i : 0.0;
                rsc[80 + lane] = (d_valid(w_i) && w_i != 0.0f) ? 1.0 / (double)w_i : 0.0;
                rR[lane] = su.fR[fi];
                L.orec[lane] = o0; L.olaf[lane] = o1.x;
            }
            const bool anyv = wave_ballot(lane < u && rsc[40 + (lane < u ? lane : 0)] != 0.0) != 0ull, anyw = wave_ballot(lane < u && rsc[80 + (lane < u ? lane : 0)] != 0.0) != 0ull;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // P(i, p) = corr(observation i, observation p) with the scales AT OBSERVATION i (oi.cpp:304-312, structure.cpp:188-214): every entry of
            // the u x u matrix, one per lane and pass
            const int nent = u * u;
            for(int e0 = 0; e0 < nent; e0 += 64) {
                const int e = min(e0 + lane, nent - 1);
                const int i = (int)(((float)e + 0.5f) / (float)u), pcol = e - i * u;
                const float4 ri = L.orec[i], rp = L.orec[pcol];
                const float hd = d_chord(ri.x, ri.y, ri.z, rp.x, rp.y, rp.z);
                float cv = d_barnes_rho_flat(hd, rsc[i]);
                if(anyv) { const float f = d_barnes_rho_flat((d_valid(ri.w) && d_valid(rp.w)) ? ri.w - rp.w : 0.0f, rsc[40 + i]); cv = (d_valid(ri.w) && d_valid(rp.w)) ? cv * f : cv; }
                if(anyw) { const float li = L.olaf[i], lp = L.olaf[pcol]; const float f = d_barnes_rho_flat((d_valid(li) && d_valid(lp)) ? li - lp : 0.0f, rsc[80 + i]); cv = (d_valid(li) && d_valid(lp)) ? cv * f : cv; }
                cv = hd > rR[i] ? 0.0f : cv;
                if(e0 + lane < nent) colbuf[pcol * U_MAXU + i] = cv;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            UPROF(7);   // P build
            // rows in lanes: the 32 register columns, the late columns 32 .. 39 (as they are: substituted behind the elimination), obs - background
            double row[NC];
#pragma unroll
            for(int p = 0; p < NC; ++p) {
                double v = 0.0;
                if(p < u && lane < u) { v = (double)colbuf[p * U_MAXU + lane]; if(lane == p) v += (double)o1.w; }     // lP + lR
                row[p] = v;
            }
            float lt[8];
#pragma unroll
            for(int b = 0; b < 8; ++b) lt[b] = (32 + b < u && lane < u) ? colbuf[(32 + b) * U_MAXU + lane] : 0.0f;
            const double dcol = lane < u ? (double)o1.y - (double)o1.z : 0.0;                                           // lObs - lY
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            auto rcp_nr = [](const double x) { double r = __builtin_amdgcn_rcp(x); r = r * (2.0 - x * r); r = r * (2.0 - x * r); return r; };
            {
                int co = 0;
#pragma unroll
                for(int j = 0; j < NC; ++j) {
                    if(j < c) {
                        const double piv = readlane_d(row[j], j);
                        if(!(fabs(piv) > 1e-290)) bad = true;     // (also NaN)
                        const double rp = rcp_nr(piv);
                        if(lane == j) {   // row j of U is final: to its place -- which is also where the rank-1 update below reads it
                            sv[oI + j] = rp;
#pragma unroll
                            for(int p = j; p < NC; ++p) if(p < u) sv[co + p - j] = row[p];
                        }
                        const double mlt = lane > j ? row[j] * rp : 0.0;     // (rows >= u hold zeros)
                        if(lane > j) row[j] = mlt;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const double* const q = sv + co - j;
#pragma unroll
                        for(int p = j + 1; p < NC; ++p) row[p] = __builtin_fma(-mlt, q[p], row[p]);     // (p >= u: registers nobody reads)
                        co += u + 1 - j;
                    }
                }
            }
            UPROF(8);   // row load + elimination
            // late columns and obs - background against L: a core lane ends with its entry of U resp. of L^-1 d, an extras lane with its entry of
            // the Schur complement resp. of d'
            const int ea = lane - c;
            const int myco = lane * (u + 1) - lane * (lane - 1) / 2;     // CO(lane)
            auto subst = [&](double t) {
#pragma unroll
                for(int j = 0; j < NC; ++j) {
                    if(j < c) { const double tj = readlane_d(t, j); t = lane > j ? __builtin_fma(-row[j], tj, t) : t; }
                }
                return t;
            };
#pragma unroll
            for(int b = 0; b < 8; ++b) {
                const int p = 32 + b;
                if(p < u) {
                    const double t = subst((double)lt[b] + (lane == p ? (double)o1.w : 0.0));
                    if(lane < c) sv[myco + p - lane] = t;
                    else if(lane < u) sv[oS + ea * nE + (p - c)] = t;
                }
            }
            {
                const double t = subst(dcol);
                if(lane < c) sv[myco + u - lane] = t;
                else if(lane < u) sv[oD + ea] = t;
            }
#pragma unroll
            for(int p = 0; p < NC; ++p) {
                if(p >= c && p < u) { if(lane >= c && lane < u) sv[oS + ea * nE + (p - c)] = row[p]; }
            }
            if(!a.allow_extrap) {   // oi.cpp:318-334: extremes of obs - background over the core (every cell selects it)
                maxInc = wave_max(lane < c ? dpf : -INFINITY);
                minInc = wave_min(lane < c ? dpf : INFINITY);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            UPROF(9);   // export
            // ---- per cell: t = U^-T g by sweeping the rows of U (row j updates every later entry at once; the entries of this cell's own extras
            //      ride along: they end as g_X - B_col^T t_C), then the LU of the cell's own block of the Schur complement
            double z[NC];
#pragma unroll
            for(int k = 0; k < NC; ++k) z[k] = (double)gf[k];
            double qx[U_MAXM];
            const double* xp[U_MAXM];
#pragma unroll
            for(int i = 0; i < U_MAXM; ++i) {
                const int ai = (elist >> (4 * i)) & 15;
                qx[i] = (i < mmax) ? (double)L.f.erho[ai][lane] : 0.0;
                xp[i] = sv + c + ai;
            }
            {
                int co = 0;
#pragma unroll
                for(int j = 0; j < NC; ++j) {
                    if(j < c) {
                        const double* const ur = sv + co - j;      // ur[k]: U(j, k); ur[u]: (L^-1 d)_j
                        const double zj = z[j] * sv[oI + j];
                        inc = __builtin_fma(zj, ur[u], inc);
#pragma unroll
                        for(int k = j + 1; k < NC; ++k) z[k] = __builtin_fma(-ur[k], zj, z[k]);
#pragma unroll
                        for(int i = 0; i < U_MAXM; ++i) if(i < mmax) { qx[i] = __builtin_fma(-*xp[i], zj, qx[i]); xp[i] += u - j; }
                        co += u + 1 - j;
                    }
                }
            }
            if(mmax > 0) {
                double lu[U_MAXM][U_MAXM], iu[U_MAXM], tq[U_MAXM], yd[U_MAXM];
#pragma unroll
                for(int i = 0; i < U_MAXM; ++i) {
                    if(i < mmax) {
                        const int ai = (elist >> (4 * i)) & 15;
                        const bool valid = i < m;
#pragma unroll
                        for(int jj = 0; jj < U_MAXM; ++jj) {
                            if(jj < mmax) {
                                const int aj = (elist >> (4 * jj)) & 15;
                                double sij = sv[oS + ai * nE + aj];
#pragma unroll
                                for(int k = 0; k < (i < jj ? i : jj); ++k) sij = __builtin_fma(-lu[i][k], lu[k][jj], sij);
                                lu[i][jj] = jj < i ? sij * iu[jj] : sij;
                            }
                        }
                        if(valid && !(fabs(lu[i][i]) > 1e-290)) bad = true;
                        iu[i] = rcp_nr(valid ? lu[i][i] : 1.0);
                        double y = sv[oD + ai], t = qx[i];
#pragma unroll
                        for(int k = 0; k < i; ++k) { y = __builtin_fma(-lu[i][k], yd[k], y); t = __builtin_fma(-lu[k][i], tq[k], t); }
                        yd[i] = y;
                        tq[i] = t * iu[i];
                        if(valid) {
                            inc = __builtin_fma(tq[i], yd[i], inc);
                            if(!a.allow_extrap) {
                                const float de = __int_as_float(LS.worig[ai]);
                                maxInc = fmaxf(maxInc, de); minInc = fminf(minInc, de);
                            }
                        }
                    }
                }
            }
        }
        else {
        if(solver) {
        // lower triangle of P (oi.cpp:304-312), one entry per lane and pass: entry e = i (i + 1) / 2 + p, p <= i
        if(lane < u) { L.orec[lane] = o0; L.olaf[lane] = o1.x; }
        float* colbuf = reinterpret_cast<float*>(L.f.solve);   // [u][U_MAXU]
        const int ntri = u * (u + 1) / 2;
        // (PLAIN: PB entries per lane and pass, their chains side by side -- chord, correctly rounded root and quotient, table exp: ~35 dependent
        //  operations each; the nine passes of a 33-row union one after the other were a wave waiting for itself, see eval_n)
#ifndef GPP_UNION_PB
#define GPP_UNION_PB 1     // (measured: 3 entries side by side 7.1 k clocks per tile, one at a time 7.2 k -- the same, for eight more registers)
#endif
        constexpr int PB = PLAIN ? GPP_UNION_PB : 1;
        const bool hh = d_valid(st.h) && st.h != 0.0f, hv = d_valid(st.v) && st.v != 0.0f, hw = d_valid(st.w) && st.w != 0.0f;
        const double rh = hh ? 1.0 / (double)st.h : 0.0, rv_ = hv ? 1.0 / (double)st.v : 0.0, rw_ = hw ? 1.0 / (double)st.w : 0.0;
        for(int e0 = 0; e0 < ntri; e0 += 64 * PB) {
            int ti[PB], tp[PB];
            float4 ri[PB], rp[PB];
            float li[PB], lp[PB];
#pragma unroll
            for(int q = 0; q < PB; ++q) {
                const int e = min(e0 + 64 * q + lane, ntri - 1);
                int i = (int)((d_sqrt_raw(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);   // (the two tests below put an estimate one off right)
                if(i * (i + 1) / 2 > e) i--;
                if((i + 1) * (i + 2) / 2 <= e) i++;
                ti[q] = i; tp[q] = e - i * (i + 1) / 2;
                ri[q] = L.orec[ti[q]]; rp[q] = L.orec[tp[q]];
                li[q] = L.olaf[ti[q]]; lp[q] = L.olaf[tp[q]];
            }
            float cv[PB];
            if constexpr(PLAIN) {
                // d_barnes_corr_flat for PB pairs, the uniform tests around the groups of chains
                float hd[PB];
#pragma unroll
                for(int q = 0; q < PB; ++q) { hd[q] = d_chord(ri[q].x, ri[q].y, ri[q].z, rp[q].x, rp[q].y, rp[q].z); cv[q] = 1.0f; }
                if(hh) {
#pragma unroll
                    for(int q = 0; q < PB; ++q) cv[q] = d_barnes_rho_flat(hd[q], rh);
                }
                if(hv) {
#pragma unroll
                    for(int q = 0; q < PB; ++q) { const float f = d_barnes_rho_flat(ri[q].w - rp[q].w, rv_); cv[q] = (d_valid(ri[q].w) && d_valid(rp[q].w)) ? cv[q] * f : cv[q]; }
                }
                if(hw) {
#pragma unroll
                    for(int q = 0; q < PB; ++q) { const float f = d_barnes_rho_flat(li[q] - lp[q], rw_); cv[q] = (d_valid(li[q]) && d_valid(lp[q])) ? cv[q] * f : cv[q]; }
                }
#pragma unroll
                for(int q = 0; q < PB; ++q) cv[q] = hd[q] > st.R ? 0.0f : cv[q];
            }
            else {
#pragma unroll
                for(int q = 0; q < PB; ++q) cv[q] = d_corr_t<PLAIN>(st, ri[q].x, ri[q].y, ri[q].z, ri[q].w, li[q], rp[q].x, rp[q].y, rp[q].z, rp[q].w, lp[q], false);
            }
#pragma unroll
            for(int q = 0; q < PB; ++q) if(e0 + 64 * q + lane < ntri) colbuf[tp[q] * U_MAXU + ti[q]] = cv[q];
        }
        // obs - background of every row (oi.cpp:293) for the row of lane 63: through the (still free) column staging area
        double* const colL = L.f.solve + (U_SOLVE - 64);   // free until the export (the P staging and 1/diag live below it)
        // (round 6: its address in a register of its own -- as "wave base + 7680 + 8 p" every pair of broadcast reads of the elimination costs an
        //  address addition, 122 of the elimination's ~1 200 vector instructions -- makes the scheduler keep more reads in flight: 168 registers
        //  and 20 bytes of scratch instead of 150 and none.  Left as it was.)
        if(lane < u) colL[lane] = (double)o1.y - (double)o1.z;                                             // lObs - lY
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        UPROF(7);   // P build
        double row[NC];
#pragma unroll
        for(int p = 0; p < NC; ++p) {
            double v = 0.0;
            if(p < u) {
                if(lane < u && lane >= p) {   // lower triangle only (all the factorisation reads)
                    v = (double)colbuf[p * U_MAXU + lane];
                    if(lane == p) v += (double)o1.w;                                                       // lP + lR
                }
                if(lane == 63) v = colL[p];
            }
            row[p] = v;
        }
        // columns 32..u-1 do not fit the 32-column register tile: their entries (rows >= 32 and the obs - background row) wait
        // in LDS and are reduced after the elimination
        const int lidx = lane == 63 ? 8 : lane - NC;
#pragma unroll
        for(int b = 0; b < (U_MAXU > NC ? 8 : 0); ++b) {
            const int p = NC + b;
            if(p < u) {
                double v = 0.0;
                if(lane < u && lane >= p) {
                    v = (double)colbuf[p * U_MAXU + lane];
                    if(lane == p) v += (double)o1.w;
                }
                if(lane == 63) v = colL[p];
                if((lane >= p && lane < u) || lane == 63) L.late[b][lidx] = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr(COLS) {
        // Right-looking Cholesky over the core columns, TWO columns per step.  The 2 x 2 pivot block [a b; b cc] gives both reciprocal roots side
        // by side -- 1/sqrt(a) and 1/sqrt(a cc - b^2) are independent chains, 1/l22 = sqrt(a) / sqrt(a cc - b^2) -- where one column at a time
        // waits for the first root, the broadcast of column j through LDS and the update of column j + 1 before the second can start: half
        // as many serial steps (the per-phase clocks of this kernel do not change when the other waves of the SIMD idle: it waits for its
        // own chains, profiles/r06_union_phases.txt).  The multiply-adds are the ones of the one-column form, in the same order.
        // An odd core ends in a step whose second column is switched off (multiplier 0; its column of the matrix -- the first extras column --
        // only takes the update of column j).
        int co = 0;   // CO(j)
#pragma unroll
        for(int j = 0; j < NC; j += 2) {
            if(j < c) {
                const bool two = j + 1 < c;
                const double pa = readlane_d(row[j], j), pb = readlane_d(row[j], j + 1), pc = readlane_d(row[j + 1], j + 1);
                const double det = two ? __builtin_fma(pa, pc, -(pb * pb)) : 1.0;
                if(!(pa > 0.0) || !(det > 0.0)) bad = true;
                const double r1 = rsqrt_nr(pa), rd = rsqrt_nr(det);
                const double l21 = pb * r1, i22 = two ? rd * (pa * r1) : 0.0;
                const double c1 = row[j] * r1;
                const double t2 = __builtin_fma(-c1, l21, row[j + 1]);
                const double c2 = t2 * i22;
                row[j] = c1;
                row[j + 1] = two ? c2 : t2;
                const int co1 = co + (u + 1 - j);   // CO(j + 1)
                if(lane == 0) { sv[oI + j] = r1; if(two) sv[oI + j + 1] = i22; }
                {   // the columns to their place: row r at CO + r - j, the obs - background row behind row u - 1
                    const int r = lane == 63 ? u : lane;
                    if((lane >= j && lane < u) || lane == 63) sv[co + r - j] = c1;
                    if(two && ((lane > j && lane < u) || lane == 63)) sv[co1 + r - (j + 1)] = c2;
                }
                // rank-2 update of the register columns behind the pair: broadcast reads (rows >= u read what lies behind the column: their
                // registers are never used; a switched-off second column reads the first one's entries again, times 0)
                const double* const q1 = sv + co - j;
                const double* const q2 = two ? sv + co1 - (j + 1) : q1;
#pragma unroll
                for(int p = j + 2; p < NC; ++p) {
                    row[p] = __builtin_fma(-c1, q1[p], row[p]);
                    row[p] = __builtin_fma(-c2, q2[p], row[p]);
                }
                co = co1 + (u - j);   // CO(j + 2)
            }
        }
        UPROF(8);   // row load + elimination
        // what is left in the register columns c .. of the extras rows and of the obs - background row: the Schur complement and d'
        const int ea = lane - c;   // extras row index of this lane
        {
            const bool erow = lane >= c && lane < u;
            const int bse = (erow ? oS + ea * nE : oD) - c;
#pragma unroll
            for(int p = 0; p < NC; ++p) {
                if(p >= c && p < u) { if((erow && p <= lane) || lane == 63) sv[bse + p] = row[p]; }
            }
        }
        if(u > 32 && !GPP_DBG(a, 16)) {   // Schur complement / d' of the late columns: entry - (row of B or L_C^-1 d) . (row p of B)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for(int b = 0; b < 8; ++b) {
                const int p = 32 + b;
                if(p < u) {
                    const bool mine = (lane >= p && lane < u) || lane == 63;
                    double acc = mine ? L.late[b][lidx] : 0.0;
                    int ck = p;   // CO(k) + p - k: entry (row p, column k)
#pragma unroll
                    for(int k = 0; k < 32; ++k) {
                        if(k < c) acc = __builtin_fma(-row[k], sv[ck], acc);
                        ck += u - k;
                    }
                    if(lane >= p && lane < u) sv[oS + ea * nE + (p - c)] = acc;
                    else if(lane == 63) sv[oD + (p - c)] = acc;
                }
            }
        }
        }
        else {
        // right-looking Cholesky over the core columns; the trailing rows/columns end as B, the Schur complement, L_C^-1 d, d'
#pragma unroll
        for(int j = 0; j < NC; ++j) {
            if(j < c) {
                const double ajj = readlane_d(row[j], j);
                if(!(ajj > 0.0)) bad = true;
                const double rs = rsqrt_nr(ajj);
                if(lane == 0) sv[oI + j] = rs;
                const double cj = row[j] * rs;
                row[j] = cj;
                // column j of the factor goes through LDS: one write, then broadcast reads (two values per ds_read) feed
                // the rank-1 update -- half the instructions of a v_readlane pair per multiply-add
                colL[lane] = cj;
#pragma unroll
                for(int p = j + 1; p < NC; ++p) row[p] = __builtin_fma(-cj, colL[p], row[p]);
            }
        }
        UPROF(8);   // row load + elimination
        // export: L_C rows packed, B rows (stride bs), L_C^-1 d, Schur complement, d'
        const int ea = lane - c;   // extras row index of this lane
        {   // one masked store per column: every lane has a last column and two bases (columns below c / from c on); written
            // with nested ifs this loop compiled to ~60 mostly scalar instructions and a dozen branches per column
            int maxp = -1, baseLo = 0, baseHi = 0;
            if(lane < c) { maxp = lane; baseLo = oL + lane * (lane + 1) / 2; baseHi = baseLo; }
            else if(lane < u) { maxp = u - 1; baseLo = oB + ea * bs; baseHi = oS + ea * nE - c; }
            else if(lane == 63) { maxp = u - 1; baseLo = oZ; baseHi = oD - c; }
#pragma unroll
            for(int p = 0; p < NC; ++p) {
                const int bse = (p < c) ? baseLo : baseHi;
                if(p <= maxp) sv[bse + p] = row[p];
            }
        }
        if(U_MAXU > NC && u > NC && !GPP_DBG(a, 16)) {   // Schur complement / d' of the late columns: entry - (row of B or L_C^-1 d) . (row p of B)
            // (row p of B has just been exported: it comes back as LDS broadcasts, one read per two multiply-adds, instead of a
            //  v_readlane pair per multiply-add; columns c.. of the padded row are multiplied by zeros)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for(int b = 0; b < 8; ++b) {
                const int p = NC + b;
                if(p < u) {
                    const bool mine = (lane >= p && lane < u) || lane == 63;
                    double acc = mine ? L.late[b][lidx] : 0.0;
                    const double* const bp = sv + oB + (p - c) * bs;
#pragma unroll
                    for(int k = 0; k < NC; ++k) if(k < c) acc = __builtin_fma(-row[k], bp[k], acc);
                    if(lane >= p && lane < u) sv[oS + ea * nE + (p - c)] = acc;
                    else if(lane == 63) sv[oD + (p - c)] = acc;
                }
            }
        }
        if(lane >= c && lane < u && (c & 1) == 0) sv[oB + ea * bs + c] = 0.0;   // padding element of the odd stride
        // The per-cell finish reads the rows of B four columns at a time: up to three doubles past a row, multiplied by z = 0.  Behind the
        // last row these are the Schur complement and d' -- and, with ONE extras row and c = 4 k + 1, the double after d', which nobody
        // has written: for c >= 57 (64-column form only) it lies outside the 16 KB of rho slots this kernel initialises, i.e. it holds
        // whatever the last workgroup on this CU left in LDS, and 0 x NaN = NaN (an all-ones sentinel key of k_oi's scan is such a pattern).
        // That was the intermittent failure of the max_points 33..62 soak (seed 5019: c = 61, one extra).
        if(lane < 3) sv[oD + nE + lane] = 0.0;
        }
        if(!a.allow_extrap) {   // oi.cpp:318-334: extremes of obs - background over the core (every cell selects it)
            maxInc = wave_max(lane < c ? dpf : -INFINITY);
            minInc = wave_min(lane < c ? dpf : INFINITY);
        }
        if constexpr(PAIR) {
            if(paired) {
                if(lane == 0) { PP->dmax = maxInc; PP->dmin = minInc; }
                if(bad && lane == 0) atomicOr(a.err, ERR_SINGULAR);   // (both tiles have cells to update)
                bad = false;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }   // solver
        if constexpr(PAIR) {
            if(paired) {
                UPROF(9);
                __syncthreads();                                                                  // B3: the factor of wave 0 is complete
                UPROF(15);  // B3 wait
                if(!solver) { maxInc = PP->dmax; minInc = PP->dmin; }
            }
        }

        UPROF(9);   // export
        // ============= per cell (lane): forward substitution of G against L_C, then its own extras ==================
        if constexpr(COLS) {
        // Column sweep: z_k = z_k / l_kk is final once the columns before k have been applied; column k then updates every later row at once --
        // independent multiply-adds behind ONE dependent pair per column, where the row form had a dot product of k terms in front of every
        // z_k.  The rows of this cell's own extras ride along (their entries of column k at per-lane addresses): they end as L_E-ready
        // right-hand sides B z, which the row form computed afterwards as one more dot product per extra.
        double z[NC];
#pragma unroll
        for(int k = 0; k < NC; ++k) z[k] = (double)gf[k];
        double qx[U_MAXM];
        const double* xp[U_MAXM];   // the entry of this cell's i-th extras row in the column walked (per lane; advanced with the column)
#pragma unroll
        for(int i = 0; i < U_MAXM; ++i) {
            const int ai = (elist >> (4 * i)) & 15;
            qx[i] = (i < mmax) ? (double)L.f.erho[ai][lane] : 0.0;
            xp[i] = sv + c + ai;
        }
        {
            int co = 0;
#pragma unroll
            for(int k = 0; k < NC; ++k) {
                if(k < c && !GPP_DBG(a, 32)) {
                    const double* const col = sv + co - k;      // col[r]: entry (row r, column k), col[u]: the obs - background row
                    const double zk = z[k] * sv[oI + k];
                    inc = __builtin_fma(zk, col[u], inc);
                    a00 = __builtin_fma(zk, zk, a00);
#pragma unroll
                    for(int i = k + 1; i < NC; ++i) z[i] = __builtin_fma(-col[i], zk, z[i]);
#pragma unroll
                    for(int i = 0; i < U_MAXM; ++i) if(i < mmax) { qx[i] = __builtin_fma(-*xp[i], zk, qx[i]); xp[i] += u - k; }
                    co += u + 1 - k;
                }
            }
        }
        if(mmax > 0 && !GPP_DBG(a, 64)) {
            double ll[U_MAXM * (U_MAXM + 1) / 2], qv[U_MAXM], tv[U_MAXM], il[U_MAXM];
#pragma unroll
            for(int i = 0; i < U_MAXM; ++i) {
                if(i < mmax) {
                    const int ai = (elist >> (4 * i)) & 15;
                    const bool valid = i < m;
                    double gq = qx[i];
                    double dq = sv[oD + ai];
                    double sii = sv[oS + ai * nE + ai];
#pragma unroll
                    for(int jj = 0; jj < i; ++jj) {
                        const int aj = (elist >> (4 * jj)) & 15;
                        double sij = sv[oS + ai * nE + aj];
#pragma unroll
                        for(int k = 0; k < jj; ++k) sij = __builtin_fma(-ll[i * (i + 1) / 2 + k], ll[jj * (jj + 1) / 2 + k], sij);
                        const double lij = sij * il[jj];
                        ll[i * (i + 1) / 2 + jj] = lij;
                        sii = __builtin_fma(-lij, lij, sii);
                        gq = __builtin_fma(-lij, qv[jj], gq);
                        dq = __builtin_fma(-lij, tv[jj], dq);
                    }
                    if(valid && !(sii > 0.0)) bad = true;
                    const double rs = rsqrt_nr(valid ? sii : 1.0);
                    il[i] = rs;
                    qv[i] = gq * rs;
                    tv[i] = dq * rs;
                    if(valid) {
                        inc = __builtin_fma(qv[i], tv[i], inc);
                        a00 = __builtin_fma(qv[i], qv[i], a00);
                        if(!a.allow_extrap) {
                            const float de = __int_as_float(LS.worig[ai]);
                            maxInc = fmaxf(maxInc, de); minInc = fminf(minInc, de);
                        }
                    }
                }
            }
        }
        }
        else {
        // ============= per cell (lane): forward substitution of G against L_C, then its own extras ==================
        double z[NC];
#pragma unroll
        for(int k = 0; k < NC; ++k) {
            double zk = 0.0;
            if(k < c && !GPP_DBG(a, 32)) {
                const double gk = (double)gf[k];
                double acc0 = 0.0, acc1 = 0.0;
                const double* lrow = sv + oL + k * (k + 1) / 2;
#pragma unroll
                for(int j = 0; j < k; ++j) {
                    if(j & 1) acc1 = __builtin_fma(lrow[j], z[j], acc1);
                    else acc0 = __builtin_fma(lrow[j], z[j], acc0);
                }
                zk = (gk - (acc0 + acc1)) * sv[oI + k];
                inc = __builtin_fma(zk, sv[oZ + k], inc);
                a00 = __builtin_fma(zk, zk, a00);
            }
            z[k] = zk;
        }
        if(mmax > 0 && !GPP_DBG(a, 64)) {
            double ll[U_MAXM * (U_MAXM + 1) / 2], qv[U_MAXM], tv[U_MAXM], il[U_MAXM];
#pragma unroll
            for(int i = 0; i < U_MAXM; ++i) {
                if(i < mmax) {
                    const int ai = (elist >> (4 * i)) & 15;
                    const bool valid = i < m;
                    const double gi = (double)L.f.erho[ai][lane];
                    const double* brow = sv + oB + ai * bs;
                    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
                    for(int j0 = 0; j0 < NC; j0 += 4) {
                        if(j0 < c) {   // reads at most 3 elements past the row (multiplied by z = 0; written data: the next row, the Schur complement, d', three zeros)
                            acc0 = __builtin_fma(brow[j0], z[j0], acc0);
                            acc1 = __builtin_fma(brow[j0 + 1], z[j0 + 1], acc1);
                            acc0 = __builtin_fma(brow[j0 + 2], z[j0 + 2], acc0);
                            acc1 = __builtin_fma(brow[j0 + 3], z[j0 + 3], acc1);
                        }
                    }
                    double gq = gi - (acc0 + acc1);
                    double dq = sv[oD + ai];
                    double sii = sv[oS + ai * nE + ai];
#pragma unroll
                    for(int jj = 0; jj < i; ++jj) {
                        const int aj = (elist >> (4 * jj)) & 15;
                        double sij = sv[oS + ai * nE + aj];
#pragma unroll
                        for(int k = 0; k < jj; ++k) sij = __builtin_fma(-ll[i * (i + 1) / 2 + k], ll[jj * (jj + 1) / 2 + k], sij);
                        const double lij = sij * il[jj];
                        ll[i * (i + 1) / 2 + jj] = lij;
                        sii = __builtin_fma(-lij, lij, sii);
                        gq = __builtin_fma(-lij, qv[jj], gq);
                        dq = __builtin_fma(-lij, tv[jj], dq);
                    }
                    if(valid && !(sii > 0.0)) bad = true;
                    const double rs = rsqrt_nr(valid ? sii : 1.0);
                    il[i] = rs;
                    qv[i] = gq * rs;
                    tv[i] = dq * rs;
                    if(valid) {
                        inc = __builtin_fma(qv[i], tv[i], inc);
                        a00 = __builtin_fma(qv[i], qv[i], a00);
                        if(!a.allow_extrap) {
                            const float de = __int_as_float(LS.worig[ai]);
                            maxInc = fmaxf(maxInc, de); minInc = fminf(minInc, de);
                        }
                    }
                }
            }
        }
        }
        }   // !SP
        UPROF(10);  // per-lane finish
        if(cnt > 0) {
            float increment = (float)inc;   // oi.cpp:317
            if(!a.allow_extrap) {           // oi.cpp:318-334
                if(maxInc > 0 && increment > maxInc) increment = maxInc;
                else if(maxInc < 0 && increment > 0) increment = maxInc;
                else if(minInc < 0 && increment < minInc) increment = minInc;
                else if(minInc > 0 && increment < 0) increment = minInc;
            }
            res_out = bg + increment;                               // oi.cpp:335
            res_var = (float)((double)bvar * (1.0 - a00));          // oi.cpp:337
        }
        if(wave_ballot(bad && cnt > 0) != 0ull && lane == 0) atomicOr(a.err, ERR_SINGULAR);
        if(lane == 0 && a.counters) {
            unsigned long long* cs = a.counters + 80 + 2 * (blockIdx.x % GPP_NSLOT);
            atomicAdd(&cs[0], (unsigned long long)__popcll(upd));
            if(solver) atomicAdd(&cs[1], 1ull);
        }
    }
#ifdef GPP_UNION_PROFILE
    if(lane < 24) atomicAdd(&a.counters[20 + 24 * pw + lane], (unsigned long long)L.prof[lane]);   // (pairs: the phases of wave 1 apart)
#endif
    if(cell >= 0) {
        a.out[cell] = res_out;
        if(a.out_var) a.out_var[cell] = res_var;
    }
}

// First pass (LIST = false): PERSISTENT waves -- a grid that just fills the chip, every wave strides over the tiles on its own.  With
// one tile per wave and four waves per workgroup the LDS of a workgroup stayed allocated until its slowest tile was done (the other
// three waves idle: tile times differ by their evictions and ring work), and the next workgroup could not start before.
// LIST = true: one item per wave, the grid sized by the host for the longest list that can arrive.
template <bool PLAIN, bool LIST, int NC, bool SP = false>   // (SP: only its list passes run through this kernel; its first pass is k_oi_union_sp)
__global__ __launch_bounds__(64 * UnionCfg<NC>::WPB, NC == 64 ? 1 : ((NC == 32 && PLAIN && !SP) ? 3 : 2)) void k_oi_union(OiArgs a) {   // the generic structure functions need more registers: never spill (see build())
    constexpr int WPB = UnionCfg<NC>::WPB;
    __shared__ UnionLds<NC> s_u[WPB];
    if constexpr(PLAIN) { d_exptab_fill<64 * WPB>(); __syncthreads(); }   // 2^(j/128) for d_exp_core
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if constexpr(LIST) {
        int tile = blockIdx.x * WPB + wid, sub = -1;   // sub: (lane >> shift) of the lanes of this item, -1 = all
        int shift = 0;
        const int nlist = *a.in_count;
        // Splitting pays while most items succeed.  Level 1: more than half of the tiles declined -> forward them whole.
        // Level 2: more than half of the 16-cell items declined (or whole tiles arrived) -> forward what arrived, unsplit.
        const bool whole = nlist > 0 && a.in_list[0] < 0;
        const bool forward = a.level != 3 && (whole || (a.level == 1 ? nlist > a.ntiles / 2 : nlist > 2 * *a.parent_count));
        if(forward) {
            for(int i = blockIdx.x * (64 * WPB) + threadIdx.x; i < nlist; i += gridDim.x * (64 * WPB)) {
                const int e = a.in_list[i];
                a.out_list[atomicAdd(a.out_count, 1)] = (a.level == 1) ? ~e : e;
            }
            return;
        }
        if(a.level == 3) {   // a short list of declined tiles goes straight to its sixteen 4-cell items: one pass instead of two
            // (launched before the host knows the length of the list, with a grid that holds a short one: a longer list is left
            //  alone here and taken by the two-level passes once the host has seen its length)
            if(16 * nlist > (int)gridDim.x * WPB || tile >= 16 * nlist) return;
            sub = tile & 15; tile = a.in_list[tile >> 4]; shift = 2;
        }
        else {
            if(tile >= 4 * nlist) return;
            const int child = tile & 3, e = a.in_list[tile >> 2];
            if(a.level == 1) { tile = e; sub = child; shift = 4; }
            else { tile = e >> 5; sub = ((e & 31) - 16) * 4 + child; shift = 2; }
        }
        union_item<PLAIN, true, NC, false, SP>(a, tile, sub, shift, s_u[wid], lane);
    }
    else if constexpr(!UnionCfg<NC>::template persistent<PLAIN>()) {   // (the other forms keep one tile per wave: the loop costs them registers they do not have)
        const int t = (int)union_xcd_order(blockIdx.x, gridDim.x) * WPB + wid, tile = a.tile0 + t;   // (tile0 / tile_n: the band of tile rows this launch covers -- all tiles, or one band of the banded host path, oi.hip)
        if(t < a.tile_n && !(a.skip_flags && a.skip_flags[tile])) union_item<PLAIN, false, NC>(a, tile, -1, 0, s_u[wid], lane);
    }
    else {
        // static part: whole rounds of the grid; dynamic tail: the remaining tiles one by one from a counter (a wave whose tiles were
        // cheap takes more of them: with static striding alone the kernel ended when the unluckiest of 3 072 waves did)
        const int stride = gridDim.x * WPB;
#if GPP_UNION_PERSIST == 2
        const int nstatic = (int)((long)a.ntiles * 7 / 8 / stride) * stride;
#else
        const int nstatic = a.ntiles;
#endif
        int tile = blockIdx.x * WPB + wid;
        for(;;) {
            if(tile >= nstatic) {
#if GPP_UNION_PERSIST == 2
                int t = 0;
                if(lane == 0) t = atomicAdd(a.tail_count, 1);
                tile = nstatic + __builtin_amdgcn_readfirstlane(t);
#endif
                if(tile >= a.ntiles) break;
            }
            if(a.skip_flags && a.skip_flags[tile]) { tile += stride; continue; }
            // (the lane number as a value the optimiser cannot see through: everything derived from it -- LDS addresses, `lane == k` masks,
            //  triangle indices -- is otherwise an invariant of this loop, computed once and kept: 15 registers over the budget of three
            //  waves per SIMD, i.e. scratch)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            union_item<PLAIN, false, NC>(a, tile, -1, 0, s_u[wid], ln);
            __builtin_amdgcn_wave_barrier();
            tile += stride;
        }
    }
}

// First pass with ONE factorisation per PAIR of neighbouring tiles (round 6; see PAIR in union_item): a workgroup is the two waves of a pair.
// Grid: tiles (2 px, 2 px + 1) of one tile row -- 8 x 16 cells on an isotropic grid; a row with an odd number of tiles ends in a single.
// Points (no 2-D tiling): chunks 2 p and 2 p + 1 of 64 consecutive points.  A wave whose tile does not exist or is left to the list passes
// (skip_flags) ends at once; its partner then runs as a single-tile item and meets no barrier.
inline long union_pair_count(const OiArgs& a) { return a.tiled2d ? (long)((a.tiles_x + 1) / 2) * (a.ntiles / a.tiles_x) : ((long)a.ntiles + 1) / 2; }
template <bool PLAIN>
__global__ __launch_bounds__(128, PLAIN ? 3 : 2) void k_oi_union_pair(OiArgs a) {
    __shared__ UnionLds<32> s_u[2];
    __shared__ PairLds s_p;
    if constexpr(PLAIN) { d_exptab_fill<128>(); __syncthreads(); }
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int t0, n;      // first tile of the pair, tiles it has (1 or 2)
    if(a.tiled2d) {
        const int px2 = (a.tiles_x + 1) >> 1;
        const int ty = (int)blockIdx.x / px2, px = (int)blockIdx.x - ty * px2;
        t0 = ty * a.tiles_x + 2 * px; n = 2 * px + 1 < a.tiles_x ? 2 : 1;
    }
    else { t0 = 2 * (int)blockIdx.x; n = t0 + 1 < a.ntiles ? 2 : 1; }
    const bool run0 = !(a.skip_flags && a.skip_flags[t0]);
    const bool run1 = n == 2 && !(a.skip_flags && a.skip_flags[t0 + 1]);
    if(!(wid == 0 ? run0 : run1)) return;
    // Which wave eliminates: the hardware places wave 0 of every workgroup on one pair of SIMDs and wave 1 on the other, so with a fixed role two
    // SIMDs of a CU would do all the factorisations while the other two wait (measured: no gain at all from the halved number of
    // factorisations).  A bit of the workgroup index that changes between the workgroups resident on a CU spreads the role.
    const int lead = (int)(__builtin_popcount(blockIdx.x * 2654435761u) & 1);
    union_item<PLAIN, false, 32, true>(a, t0 + wid, -1, 0, s_u[wid], lane, run0 && run1 && !a.pair_solo, wid, lead, s_u, &s_p);
}

// First pass for a spatially varying Barnes structure (round 6; see SP in union_item): one LU per tile.  What it declines goes through the list
// passes (k_oi_union<true, true, 32, true>: 16-cell and 4-cell items) and, what those decline, to k_oi (pivoted LU per distinct selection).
template <int V>      // (a template only so that the header can be included by two translation units)
__global__ __launch_bounds__(128, 3) void k_oi_union_sp(OiArgs a) {
    __shared__ UnionLds<32> s_u[2];
    d_exptab_fill<128>();
    __syncthreads();
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = (int)union_xcd_order(blockIdx.x, gridDim.x) * 2 + wid;
    if(t < a.tile_n) union_item<true, false, 32, false, true>(a, a.tile0 + t, -1, 0, s_u[wid], lane);
}

// grid of the persistent first pass: as many workgroups as the chip holds of this kernel at once (asked once per kernel), at most one per WPB tiles
template <void (*K)(OiArgs)>
inline unsigned union_persist_grid(const int threads, const long nb) {
    static int resident = 0;
    if(resident == 0) {
        int dev = 0, cus = 0, per_cu = 0;
        GPP_HIP(hipGetDevice(&dev));
        GPP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        GPP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(K), threads, 0));
        resident = std::max(1, cus * std::max(1, per_cu));
    }
    return (unsigned)std::min<long>(std::max<long>(nb, 1), (long)resident);
}
