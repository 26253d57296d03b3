"""CPU restatement of the reference's retrieval evaluation -- TEST INFRASTRUCTURE ONLY (imported by tests/ only).

evaluate.py:162-206 `get_recall(m, n, DATABASE_VECTORS, QUERY_VECTORS, QUERY_SETS)`: for every query of run n that has true
neighbours in run m, the 25 nearest database descriptors of run m (sklearn KDTree, Euclidean); recall@N counts the first
rank at which a true neighbour appears (cumulative, in percent), the top-1 similarity is the dot product with the first
hit when it is rank 0, and the one-percent recall asks whether the first max(round(len(db)/100), 1) ranks contain a true
neighbour.  Two forms: `get_recall_kdtree` calls the same library routine the reference calls; `get_recall_bruteforce` is
the plain numpy form the GPU path is compared against (identical results whenever no two database descriptors are
equidistant from a query at fp64 resolution).
"""
import numpy as np

RECALL_NUM = 25   # evaluate.py:19 (global `recall_num`)


def _score(db, queries, true_sets, knn_fn, recall_num):
    hits_at = np.zeros(recall_num)
    top1_similarity = []
    one_percent_hits = 0
    threshold = max(int(round(len(db) / 100.0)), 1)
    evaluated = 0
    for i, q in enumerate(queries):
        truth = true_sets[i]
        if len(truth) == 0:
            continue
        evaluated += 1
        ranked = knn_fn(q)
        truth_set = set(int(t) for t in truth)
        for rank, j in enumerate(ranked):
            if int(j) in truth_set:
                if rank == 0:
                    top1_similarity.append(float(np.dot(q, db[j])))
                hits_at[rank] += 1
                break
        if truth_set.intersection(int(j) for j in ranked[:threshold]):
            one_percent_hits += 1
    one_percent_recall = one_percent_hits / float(evaluated) * 100
    recall = np.cumsum(hits_at) / float(evaluated) * 100
    return recall, top1_similarity, one_percent_recall


def get_recall_kdtree(m, n, database_vectors, query_vectors, query_sets, recall_num=RECALL_NUM):
    from sklearn.neighbors import KDTree
    db, queries = database_vectors[m], query_vectors[n]
    tree = KDTree(db)
    true_sets = [query_sets[n][i][m] for i in range(len(queries))]
    return _score(db, queries, true_sets, lambda q: tree.query(np.array([q]), k=recall_num)[1][0], recall_num)


def get_recall_bruteforce(m, n, database_vectors, query_vectors, query_sets, recall_num=RECALL_NUM):
    db, queries = np.asarray(database_vectors[m], np.float64), query_vectors[n]
    true_sets = [query_sets[n][i][m] for i in range(len(queries))]

    def knn(q):
        d2 = ((db - np.asarray(q, np.float64)) ** 2).sum(1)
        return np.lexsort((np.arange(len(db)), d2))[:recall_num]
    return _score(database_vectors[m], queries, true_sets, knn, recall_num)


def synthetic_runs(seed, runs=3, per_run=(120, 90, 150), dim=256):
    """Synthetic DATABASE_VECTORS / QUERY_VECTORS / QUERY_SETS with the reference's structure: unit descriptors of `runs`
    traversals of the same places (a shared latent place vector + noise), true neighbours = same or adjacent place."""
    g = np.random.default_rng(seed)
    places = g.standard_normal((max(per_run), dim))
    vecs, place_of = [], []
    for r in range(runs):
        ids = np.sort(g.choice(len(places), size=per_run[r], replace=False))
        v = places[ids] + 0.6 * g.standard_normal((per_run[r], dim))
        vecs.append((v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32))
        place_of.append(ids)
    query_sets = []
    for n in range(runs):
        run = []
        for i in range(per_run[n]):
            entry = {}
            for m in range(runs):
                near = np.nonzero(np.abs(place_of[m] - place_of[n][i]) <= 1)[0]
                entry[m] = near.tolist() if g.random() > 0.1 else []      # some queries have no ground truth in a run
            run.append(entry)
        query_sets.append(run)
    return vecs, vecs, query_sets
