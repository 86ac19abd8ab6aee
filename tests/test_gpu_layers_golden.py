"""The HIP layers against what the REFERENCE'S OWN LAYER TEXT computes (tests/golden/g6_layers_*.npz: /root/reference/kgcn/
layers.py executed over the numpy stand-in of tests/golden/tf_standin.py, float64; see tests/test_oracle_layers_golden.py for
what that evidence is and is not).  Same reference-API calls as a kGCN model file makes: layer(x, adj=adjs) with adjs[b][ch] COO
triples, 10 real + 20 dummy graphs, plain / Kipf-normalised / degree-split (6 channels incl. the dummy [0, 0] -> 0.0 entries)."""
import numpy as np
import pytest
import torch

from test_gpu_parity import close, dev, t32
from test_oracle_layers_golden import TAGS, g6

pytestmark = pytest.mark.gpu


def _set(params, values):
    with torch.no_grad():
        for p, v in zip(params, values):
            p.copy_(t32(np.asarray(v).reshape(tuple(p.shape))))


@pytest.mark.parametrize("variant", ["default", "bspmm", "bconv", "batched"])
@pytest.mark.parametrize("tag", TAGS)
def test_graphconv_matches_the_reference_layer_text(tag, variant):
    from kgcn_amd import layers
    from test_gpu_parity import _set_variant
    z, adjs, C = g6(tag)
    try:
        _set_variant(variant)
        layer = layers.GraphConv(9, C)
        tx = t32(z["x"])
        layer(tx, adj=adjs)
        _set(layer.w, z["conv_w"]); _set(layer.bias, z["conv_b"])
        close(layer(tx, adj=adjs), z["conv_out"], atol=1e-5, what="GraphConv %s vs reference layer text" % variant)
    finally:
        _set_variant("default")


@pytest.mark.parametrize("tag", TAGS)
def test_gin_gather_maxpool_gat_match_the_reference_layer_text(tag):
    from kgcn_amd import layers
    z, adjs, C = g6(tag)
    tx = t32(z["x"])
    gin = layers.GINAggregate(C)
    gin(tx, adj=adjs)
    _set(gin.epsilon, z["gin_eps"])
    close(gin(tx, adj=adjs), z["gin_out"], atol=1e-5, what="GINAggregate (default branch keeps eps x, Q1)")
    close(layers.GraphGather()(t32(z["conv_out"])), z["gather_out"], atol=1e-5, what="GraphGather (padded rows summed, Q4)")
    close(layers.GraphMaxPooling(C)(tx, adj=adjs), z["maxpool_out"], atol=1e-5, what="GraphMaxPooling")
    gat = layers.GAT(C).to(dev())
    gat(tx, adj=adjs)
    _set(gat.weight_a, z["gat_a"])
    close(gat(tx, adj=adjs), z["gat_out"], atol=1e-5, what="GAT")


@pytest.mark.parametrize("tag", TAGS)
def test_graphdense_and_batchnorm_match_the_reference_layer_text(tag):
    from kgcn_amd import layers
    z, _, _ = g6(tag)
    d = layers.GraphDense(6)
    tx = t32(z["x"])
    d(tx)
    _set([d.kernel, d.bias], [z["dense_k"], z["dense_b"]])
    close(d(tx), z["dense_out"], atol=1e-5, what="GraphDense")
    tr = t32(z["ragged_x"])
    close(d(tr, enabled_node_nums=z["ragged_sizes"]), z["ragged_dense_out"], atol=1e-5, what="ragged GraphDense (zero padding, no bias)")
    bn = layers.GraphBatchNormalization()
    close(bn(tr, max_node_num=10), z["bn_out"], atol=1e-5, what="GraphBatchNormalization (learning phase 0)")
    bnr = layers.GraphBatchNormalization()
    close(bnr(tr, enabled_node_nums=z["ragged_sizes"], max_node_num=10), z["bn_ragged_out"], atol=1e-5, what="ragged GraphBatchNormalization")
