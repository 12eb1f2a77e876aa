from ex import *
ec = load("e.coli-EC590.fasta.gz"); k12 = load("e.coli-K12.fasta.gz")
for hv in (0,1):
  for mm in (0,1):
    s_ec, m_ec = sketch(ec, hashvar=hv, marker_mode=mm); s_k, m_k = sketch(k12, hashvar=hv, marker_mode=mm)
    ue=np.unique(s_ec['kmer']); uk=np.unique(s_k['kmer'])
    print(hv, mm, "seeds", len(s_ec), len(s_k), "uniq", len(ue), len(uk), "shared", len(np.intersect1d(ue,uk)), "markers", len(m_ec), len(m_k), "shared markers", len(np.intersect1d(m_ec, m_k)))
