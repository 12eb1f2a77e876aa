from ex import *
ec = load("e.coli-EC590.fasta.gz"); k12 = load("e.coli-K12.fasta.gz")
s_ec, m_ec = sketch(ec); s_k, m_k = sketch(k12)
print(m_ec[:5], m_k[:5])
# find unique shared seeds and compare 21-windows
ue,ie=np.unique(s_ec['kmer'],return_index=True); uk,ik=np.unique(s_k['kmer'],return_index=True)
sh,a,b=np.intersect1d(ue,uk,return_indices=True)
same=0
for x,y in list(zip(ie[a],ik[b]))[:2000]:
    pe=s_ec['pos'][x]; pk=s_k['pos'][y]
    we=ec[pe-20:pe+1]; wk=k12[pk-20:pk+1]
    same += (we==wk)
print(same)
