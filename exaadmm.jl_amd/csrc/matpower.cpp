// Native MATPOWER ingest -> exaadmm_grid.
// Behaviour of the reference's loader, which this replaces on the way into the hot path:
//   /root/reference/src/utils/parse_matpower.jl:1-16,46-88  section scan, rows until "];", '%' comments
//   /root/reference/src/utils/parse_matpower.jl:120-160     status != 1 generators / branches dropped, per-unit
//   /root/reference/src/utils/opfdata.jl:158-164,199-206,220  single reference bus, Inf -> +-999.99, quadratic cost
//   /root/reference/src/utils/opfdata.jl:417-456            admittances and shunts
//   /root/reference/src/utils/opfdata.jl:610-618,694-714,817-822  CSR maps, bounds, rateA rule, brBusIdx
#include "exaadmm_hip.h"

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

struct exaadmm_case {
    double baseMVA = 100.0;
    int bus_ref = -1;
    std::vector<double> pgmin, pgmax, qgmin, qgmax, c2, c1, c0;
    std::vector<double> Y[8];  // YffR YffI YftR YftI YttR YttI YtfR YtfI
    std::vector<double> FrVm, ToVm, FrVa, ToVa, rateA;
    std::vector<int64_t> brBus;
    std::vector<double> Pd, Qd, Vmin, Vmax, YshR, YshI;
    std::vector<int64_t> FrStart, FrIdx, ToStart, ToIdx, GenStart, GenIdx;
};

namespace {

thread_local std::string g_io_err;

std::string trim(const std::string& s)
{
    size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    size_t b = s.find_last_not_of(" \t\r\n");
    return s.substr(a, b - a + 1);
}

double to_num(std::string t)
{
    while (!t.empty() && (t.back() == ';' || t.back() == ',')) t.pop_back();
    if (t == "Inf" || t == "inf" || t == "+Inf") return std::numeric_limits<double>::infinity();
    if (t == "-Inf" || t == "-inf") return -std::numeric_limits<double>::infinity();
    return strtod(t.c_str(), nullptr);
}

typedef std::vector<std::vector<double>> Table;

bool read_sections(const char* path, int gen_cols, double& baseMVA, Table& bus, Table& gen, Table& branch, Table& gencost)
{
    std::ifstream in(path);
    if (!in) {
        g_io_err = std::string("cannot open ") + path;
        return false;
    }
    std::vector<std::string> lines;
    for (std::string ln; std::getline(in, ln);) lines.push_back(ln);
    bool have_base = false;
    for (size_t p = 0; p < lines.size(); p++) {
        std::istringstream is(lines[p]);
        std::string key;
        is >> key;
        Table* tab = nullptr;
        size_t ncol = 0;
        if (key == "mpc.baseMVA") {
            size_t eq = lines[p].find('=');
            if (eq != std::string::npos) {
                baseMVA = to_num(trim(lines[p].substr(eq + 1)));
                have_base = true;
            }
            continue;
        } else if (key == "mpc.bus") {
            tab = &bus; ncol = 13;
        } else if (key == "mpc.gen") {
            tab = &gen; ncol = (size_t)gen_cols;
        } else if (key == "mpc.branch") {
            tab = &branch; ncol = 13;
        } else if (key == "mpc.gencost") {
            tab = &gencost; ncol = 7;
        } else {
            continue;
        }
        for (p++; p < lines.size(); p++) {
            std::string body = lines[p];
            if (trim(body).rfind("];", 0) == 0) break;
            size_t pc = body.find('%');
            if (pc != std::string::npos) body = body.substr(0, pc);
            std::istringstream rs(body);
            std::vector<double> row;
            for (std::string tok; rs >> tok && row.size() < ncol;) row.push_back(to_num(tok));
            if (row.empty()) continue;
            if (row.size() < ncol) {
                g_io_err = "short row in section " + key;
                return false;
            }
            tab->push_back(row);
        }
    }
    if (!have_base || bus.empty() || branch.empty()) {
        g_io_err = "missing mpc.baseMVA / mpc.bus / mpc.branch";
        return false;
    }
    return true;
}

void build_csr(const std::vector<int>& owner, int nbus, std::vector<int64_t>& start, std::vector<int64_t>& idx)
{
    start.assign(nbus + 1, 0);
    for (int o : owner) start[o + 1]++;
    for (int i = 0; i < nbus; i++) start[i + 1] += start[i];
    idx.assign(owner.size(), 0);
    std::vector<int64_t> fill(start.begin(), start.end() - 1);
    for (size_t k = 0; k < owner.size(); k++) idx[fill[owner[k]]++] = (int64_t)k;  // stable: component order within a bus
}

}  // namespace

extern "C" const char* exaadmm_io_error(void) { return g_io_err.c_str(); }

extern "C" int exaadmm_load_matpower(const char* path, double tight_factor, const char* case_format, exaadmm_case** out)
{
    if (!path || !out) return EXAADMM_ERR_INVALID;
    const bool pglib = case_format && strcmp(case_format, "pglib") == 0;
    double M = 100.0;
    Table bus, gen, br, gc;
    if (!read_sections(path, pglib ? 10 : 21, M, bus, gen, br, gc)) return EXAADMM_ERR_IO;
    if (gc.size() < gen.size()) {
        g_io_err = "mpc.gencost has fewer rows than mpc.gen";
        return EXAADMM_ERR_IO;
    }
    exaadmm_case* c = new exaadmm_case();
    c->baseMVA = M;
    const int nbus = (int)bus.size();
    std::unordered_map<long long, int> id2row;
    for (int i = 0; i < nbus; i++) {
        id2row[(long long)bus[i][0]] = i;
        if ((int)bus[i][1] == 3) {
            if (c->bus_ref >= 0) {
                g_io_err = "more than one reference bus";
                delete c;
                return EXAADMM_ERR_IO;
            }
            c->bus_ref = i;
        }
        c->Pd.push_back(bus[i][2]);
        c->Qd.push_back(bus[i][3]);
        c->YshR.push_back(bus[i][4] / M);
        c->YshI.push_back(bus[i][5] / M);
        c->Vmax.push_back(bus[i][11]);
        c->Vmin.push_back(bus[i][12]);
    }
    if (c->bus_ref < 0) {
        g_io_err = "reference bus not found";
        delete c;
        return EXAADMM_ERR_IO;
    }
    auto cap = [](double v, double lim) { return std::isinf(v) ? lim : v; };
    std::vector<int> gen_bus;
    for (size_t g = 0; g < gen.size(); g++) {
        if ((int)gen[g][7] != 1) continue;
        if (!((int)gc[g][0] == 2 && (int)gc[g][3] == 3)) {
            g_io_err = "only quadratic polynomial generator costs are supported";
            delete c;
            return EXAADMM_ERR_IO;
        }
        auto itb = id2row.find((long long)gen[g][0]);
        if (itb == id2row.end()) {
            g_io_err = "generator on unknown bus";
            delete c;
            return EXAADMM_ERR_IO;
        }
        gen_bus.push_back(itb->second);
        c->pgmax.push_back(cap(gen[g][8] / M, 999.99));
        c->pgmin.push_back(cap(gen[g][9] / M, -999.99));
        c->qgmax.push_back(cap(gen[g][3] / M, 999.99));
        c->qgmin.push_back(cap(gen[g][4] / M, -999.99));
        c->c2.push_back(gc[g][4]);
        c->c1.push_back(gc[g][5]);
        c->c0.push_back(gc[g][6]);
    }
    std::vector<int> fr_bus, to_bus;
    const double two_pi = 2.0 * M_PI;
    for (size_t l = 0; l < br.size(); l++) {
        if ((int)br[l][10] != 1) continue;
        auto f = id2row.find((long long)br[l][0]), t = id2row.find((long long)br[l][1]);
        if (f == id2row.end() || t == id2row.end()) {
            g_io_err = "branch on unknown bus";
            delete c;
            return EXAADMM_ERR_IO;
        }
        const int fr = f->second, to = t->second;
        fr_bus.push_back(fr);
        to_bus.push_back(to);
        typedef std::complex<double> cd;
        const cd Ys = 1.0 / cd(br[l][2], br[l][3]);
        double ratio = br[l][8] == 0.0 ? 1.0 : br[l][8];
        const cd tap = ratio * std::exp(cd(0.0, br[l][9] * M_PI / 180.0));
        const cd Ytt = Ys + cd(0.0, br[l][4] / 2);
        const cd Yff = Ytt / (tap * std::conj(tap));
        const cd Yft = -Ys / std::conj(tap);
        const cd Ytf = -Ys / tap;
        const cd ys[4] = {Yff, Yft, Ytt, Ytf};
        for (int k = 0; k < 4; k++) {
            c->Y[2 * k].push_back(ys[k].real());
            c->Y[2 * k + 1].push_back(ys[k].imag());
        }
        c->FrVm.push_back(c->Vmin[fr]);
        c->FrVm.push_back(c->Vmax[fr]);
        c->ToVm.push_back(c->Vmin[to]);
        c->ToVm.push_back(c->Vmax[to]);
        const bool fref = (fr == c->bus_ref), tref = (to == c->bus_ref);
        c->FrVa.push_back(fref ? 0.0 : -two_pi);
        c->FrVa.push_back(fref ? 0.0 : two_pi);
        c->ToVa.push_back(tref ? 0.0 : -two_pi);
        c->ToVa.push_back(tref ? 0.0 : two_pi);
        const double ra = br[l][5];
        c->rateA.push_back(ra == 0.0 ? 1e3 : tight_factor * (ra / M) * (ra / M));
        c->brBus.push_back(fr);
        c->brBus.push_back(to);
    }
    build_csr(fr_bus, nbus, c->FrStart, c->FrIdx);
    build_csr(to_bus, nbus, c->ToStart, c->ToIdx);
    build_csr(gen_bus, nbus, c->GenStart, c->GenIdx);
    *out = c;
    return EXAADMM_OK;
}

extern "C" int exaadmm_case_grid(const exaadmm_case* c, exaadmm_grid* g)
{
    if (!c || !g) return EXAADMM_ERR_INVALID;
    memset(g, 0, sizeof(*g));
    g->ngen = (int64_t)c->pgmin.size();
    g->nline = (int64_t)c->rateA.size();
    g->nbus = (int64_t)c->Pd.size();
    g->index_base = 0;
    g->baseMVA = c->baseMVA;
    g->pgmin = c->pgmin.data(); g->pgmax = c->pgmax.data();
    g->qgmin = c->qgmin.data(); g->qgmax = c->qgmax.data();
    g->c2 = c->c2.data(); g->c1 = c->c1.data(); g->c0 = c->c0.data();
    g->YffR = c->Y[0].data(); g->YffI = c->Y[1].data();
    g->YftR = c->Y[2].data(); g->YftI = c->Y[3].data();
    g->YttR = c->Y[4].data(); g->YttI = c->Y[5].data();
    g->YtfR = c->Y[6].data(); g->YtfI = c->Y[7].data();
    g->FrVmBound = c->FrVm.data(); g->ToVmBound = c->ToVm.data();
    g->FrVaBound = c->FrVa.data(); g->ToVaBound = c->ToVa.data();
    g->rateA = c->rateA.data();
    g->brBusIdx = c->brBus.data();
    g->Pd = c->Pd.data(); g->Qd = c->Qd.data();
    g->Vmin = c->Vmin.data(); g->Vmax = c->Vmax.data();
    g->YshR = c->YshR.data(); g->YshI = c->YshI.data();
    g->FrStart = c->FrStart.data(); g->FrIdx = c->FrIdx.data();
    g->ToStart = c->ToStart.data(); g->ToIdx = c->ToIdx.data();
    g->GenStart = c->GenStart.data(); g->GenIdx = c->GenIdx.data();
    g->nline_owned = 0;
    return EXAADMM_OK;
}

extern "C" int exaadmm_case_free(exaadmm_case* c)
{
    delete c;
    return EXAADMM_OK;
}
