// Rate of psk_query_host from N host threads WITHOUT Python (what the library itself sustains; profiles/scripts/query_threads8.py is the same
// workload through pyskani_amd.Database): 10 families x 100 references of 2 Mb, c = 30 / marker_c = 200, contigs of 2-50 kb (log-uniform).
// build: g++ -O2 -std=c++17 -I include profiles/micro/query_threads.cpp -o /tmp/query_threads -L pyskani_amd -lpyskani_amd -Wl,-rpath,$PWD/pyskani_amd -lpthread
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <thread>
#include <vector>
#include "pyskani_amd.h"

static std::string mutate(std::mt19937_64& g, const std::string& a, size_t st, size_t len, double d) {
    std::string s = a.substr(st, len);
    std::uniform_real_distribution<double> u(0, 1);
    for (auto& c : s) if (u(g) < d) c = "ACGT"[g() & 3];
    return s;
}
#define CK(x) do { psk_status _s = (x); if (_s != PSK_OK) { fprintf(stderr, "%s: %s\n", #x, psk_last_error()); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int NQ = argc > 1 ? atoi(argv[1]) : 4000;
    std::mt19937_64 g(1);
    const int n_fam = 10, per = 100;
    std::vector<std::string> anc(n_fam);
    for (auto& a : anc) { a.resize(2000000); for (auto& c : a) c = "ACGT"[g() & 3]; }
    psk_ctx* ctx; CK(psk_ctx_create(0, &ctx));
    psk_params prm{30, 200, 15};
    psk_db* db; CK(psk_db_create(ctx, &prm, &db));
    {
        std::vector<std::string> refs; std::vector<const uint8_t*> ptr; std::vector<uint64_t> len; std::vector<uint32_t> gfc{0};
        for (int f = 0; f < n_fam; f++) for (int j = 0; j < per; j++) refs.push_back(mutate(g, anc[f], 0, anc[f].size(), 0.001 * j));
        for (auto& r : refs) { ptr.push_back((const uint8_t*)r.data()); len.push_back(r.size()); gfc.push_back((uint32_t)ptr.size()); }
        std::vector<psk_sketch*> sk(refs.size());
        CK(psk_sketch_many_host(ctx, &prm, ptr.data(), len.data(), gfc.data(), (uint32_t)refs.size(), 1, sk.data()));
        for (size_t i = 0; i < sk.size(); i++) { std::string nm = "r" + std::to_string(i); CK(psk_db_add(db, nm.c_str(), sk[i])); }
    }
    std::vector<std::string> q(NQ);
    std::uniform_real_distribution<double> u(0, 1);
    for (int i = 0; i < NQ; i++) {
        const std::string& a = anc[i % n_fam];
        const size_t L = (size_t)std::exp(std::log(2000.0) + u(g) * (std::log(50000.0) - std::log(2000.0)));
        q[i] = mutate(g, a, (size_t)(u(g) * (a.size() - L)), L, 0.05 * u(g));
    }
    psk_query_opts o{}; o.learned_ani = 0;
    auto run = [&](int nt) {
        std::vector<std::thread> th; std::vector<uint64_t> hits(nt, 0);
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < nt; k++) th.emplace_back([&, k] {
            for (int i = k * NQ / nt; i < (k + 1) * NQ / nt; i++) {
                const uint8_t* p = (const uint8_t*)q[i].data(); uint64_t l = q[i].size(); psk_hit* h; uint64_t n;
                CK(psk_query_host(db, &p, &l, 1, 1, &o, &h, &n));
                hits[k] += n; psk_free(h);
            }
        });
        for (auto& t : th) t.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint64_t tot = 0; for (auto x : hits) tot += x;
        printf("%2d threads: %8.0f queries/s (%llu hits)\n", nt, NQ / s, (unsigned long long)tot); fflush(stdout);
    };
    run(1);                  // warm-up: the device tables of the database
    for (int nt : {1, 2, 4, 8, 16}) run(nt);
    uint64_t t, r, gen; psk_ctx_small_query_stats(ctx, &t, &r, &gen);
    printf("one launch sequence: %llu, rerun: %llu, general path: %llu\n", (unsigned long long)t, (unsigned long long)r, (unsigned long long)gen);
    psk_db_destroy(db); psk_ctx_destroy(ctx);
    return 0;
}
