from ex import *
ec = load("e.coli-EC590.fasta.gz"); k12 = load("e.coli-K12.fasta.gz")
s_ec, m_ec = sketch(ec); s_k, m_k = sketch(k12)
iv, A, ch = chain(s_k, s_ec)
print("rev frac", A['rev'].mean())
s_ec1, m_ec1 = sketch(ec, marker_mode=1, mc=125); s_k1, m_k1 = sketch(k12, marker_mode=1,mc=125)
print(len(m_ec1), len(m_k1), len(np.intersect1d(m_ec1,m_k1)))
