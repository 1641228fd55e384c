import time, torch, sys
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
from clane_amd import synth, _hip
from clane_amd.graph import Graph
from clane_amd.embedder import Embedder
from clane_amd.similarity import CosineSimilarity
dev = _hip.require_gpu("cuda:0")
csr = synth.uniform_csr(2708, 5429, seed=0)
X = synth.bow_X(2708, 1433, seed=0)
for rep in range(2):
    g = Graph.from_csr(csr, X)
    emb = Embedder(g, CosineSimilarity(), dev, gamma=0.76, tolerence=10, verbose=False)
    torch.cuda.synchronize(); t = time.perf_counter()
    emb.iterate()
    torch.cuda.synchronize(); w = time.perf_counter() - t
    print("cora-shape iterate", round(w, 3), "s; rounds", len(emb.sweep_counts), "sweeps", sum(emb.sweep_counts), "launched", emb.sweeps_launched,
          "us per launched sweep", round(w / max(emb.sweeps_launched, 1) * 1e6, 1))
