"""Model assemblies that call the layers exactly the way the reference's models.py does
(torch_rgcn/models.py:137-200 NodeClassifier, :248-296 EmbeddingNodeClassifier,
:14-134 LinkPredictor, :208-245 CompressionRelationPredictor).

They exist so that experiments written against `torch_rgcn.models` run unchanged on
the HIP layers.  Three upstream defects are NOT reproduced: the
one-argument call of the schlichtkrull initialiser, the debug print + exit() in
LinkPredictor.forward, and c-rgcn feeding an nhid-wide tensor to an nemb-wide layer.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .layers import DistMult, RelationalGraphConvolutionLP, RelationalGraphConvolutionNC
from .utils import add_inverse_and_self, schlichtkrull_normal_, select_w_init


class NodeClassifier(nn.Module):
    """Featureless (or featured) RGC layer -> ReLU -> RGC layer producing class scores."""

    def __init__(self, triples=None, nnodes=None, nrel=None, nfeat=None, nhid=16, nlayers=2, nclass=None,
                 edge_dropout=None, decomposition=None, nemb=None):
        super().__init__()
        assert (triples is not None or nnodes is not None or nrel is not None or nclass is not None), \
            "The following must be specified: triples, number of nodes, number of relations and number of classes!"
        assert 0 < nlayers < 3, "Only supports the following number of RGCN layers: 1 and 2."
        self.nlayers = nlayers
        if nlayers == 1:
            nhid = nclass
        else:
            assert nhid is not None, "Number of hidden layers not specified!"
        base = torch.as_tensor(triples, dtype=torch.long)
        self.register_buffer('triples', base)
        self.register_buffer('triples_plus', add_inverse_and_self(base, nnodes, nrel))
        common = dict(triples=self.triples_plus, num_nodes=nnodes, num_relations=2 * nrel + 1,
                      edge_dropout=edge_dropout, decomposition=decomposition)
        self.rgc1 = RelationalGraphConvolutionNC(in_features=nfeat, out_features=nhid, vertical_stacking=False, **common)
        if nlayers == 2:
            self.rgc2 = RelationalGraphConvolutionNC(in_features=nhid, out_features=nclass, vertical_stacking=True,
                                                     **common)

    def forward(self):
        if self.nlayers == 2:      # ReLU in the first layer's epilogue where its kernel has one (reference: F.relu(self.rgc1()))
            return self.rgc2(features=self.rgc1.forward_activated(None, "relu", private=True))
        return self.rgc1()


class EmbeddingNodeClassifier(NodeClassifier):
    """e-rgcn: learnt node embeddings -> diagonal-weight RGC layer -> ReLU -> RGC layer."""

    def __init__(self, triples=None, nnodes=None, nrel=None, nfeat=None, nhid=16, nlayers=2, nclass=None,
                 edge_dropout=None, decomposition=None, nemb=None):
        assert nemb is not None, "Size of node embedding not specified!"
        assert nlayers == 2, "For this model only 2 layers are normally configured (for now)"
        super().__init__(triples, nnodes, nrel, nemb, nemb, 1, nclass, edge_dropout, decomposition)
        self.rgcn_no_hidden = RelationalGraphConvolutionNC(
            triples=self.triples_plus, num_nodes=nnodes, num_relations=2 * nrel + 1, in_features=nemb,
            out_features=nemb, edge_dropout=edge_dropout, decomposition=decomposition, vertical_stacking=False,
            diag_weight_matrix=True)
        self.node_embeddings = nn.Parameter(torch.empty(nnodes, nemb))
        nn.init.kaiming_normal_(self.node_embeddings, mode='fan_in')

    def forward(self):
        return self.rgc1(features=self.rgcn_no_hidden.forward_activated(self.node_embeddings, "relu", private=True))


def _init_embedding(tensor, name, gain=1.0):
    init = select_w_init(name)
    if init is schlichtkrull_normal_:
        init(tensor, shape=tensor.shape, gain=gain)  # upstream omits `shape` and raises TypeError
    else:
        init(tensor)


class LinkPredictor(nn.Module):
    """RGCN encoder (1-2 LP layers over learnt embeddings) + DistMult decoder."""

    def __init__(self, nnodes=None, nrel=None, nfeat=None, encoder_config=None, decoder_config=None):
        super().__init__()
        enc, dec = encoder_config or {}, decoder_config or {}
        nemb, nhid1, nhid2 = enc.get("node_embedding"), enc.get("hidden1_size"), enc.get("hidden2_size")
        layers = enc.get("num_layers", 2)
        assert (nnodes is not None or nrel is not None or nhid1 is not None), \
            "The following must be specified: number of nodes, number of relations and output dimension!"
        assert 0 < layers < 3, "Only supports the following number of convolution layers: 1 and 2."
        self.num_nodes, self.num_rels, self.rgcn_layers, self.nemb = nnodes, nrel, layers, nemb
        self.decoder_l2_type, self.decoder_l2 = dec.get("l2_penalty_type"), dec.get("l2_penalty")

        self.node_embeddings = nn.Parameter(torch.empty(nnodes, nemb))
        self.node_embeddings_bias = nn.Parameter(torch.zeros(1, nemb))
        _init_embedding(self.node_embeddings, enc.get("weight_init", "glorot-normal"))

        lp = dict(num_nodes=nnodes, num_relations=2 * nrel + 1, edge_dropout=enc.get("edge_dropout"),
                  decomposition=enc.get("decomposition"), vertical_stacking=False,
                  w_init=self._layer_init(enc.get("weight_init")), w_gain=enc.get("include_gain", False),
                  b_init=enc.get("bias_init"))
        self.rgc1 = RelationalGraphConvolutionLP(in_features=self._rgc1_in(nemb, nhid1), out_features=nhid1, **lp)
        if layers == 2:
            self.rgc2 = RelationalGraphConvolutionLP(in_features=nhid1, out_features=nhid2, **lp)
        self.scoring_function = DistMult(nrel, nemb, nnodes, nrel, self._layer_init(dec.get("weight_init"), "standard-normal"),
                                         dec.get("include_gain", False), dec.get("bias_init"))

    @staticmethod
    def _layer_init(name, default="glorot-normal"):
        # the LP layer / DistMult call init(tensor, gain=...) -- schlichtkrull needs a shape there and upstream crashes
        # (SURVEY F4a); every shipped LP config asks for it, so the substitution is said out loud, once
        if name is not None and name.lower().startswith("schlichtkrull"):
            import warnings
            warnings.warn(f"weight_init '{name}' cannot initialise the R-GCN layer / decoder weights (upstream passes no shape and "
                          f"fails there): using '{default}' for them; node embeddings keep '{name}'", stacklevel=3)
            return default
        return default if name is None else name

    @staticmethod
    def _rgc1_in(nemb, nhid1):
        return nemb

    def compute_penalty(self, batch, x):
        if self.decoder_l2 == 0.0:
            return 0
        if self.decoder_l2_type == 'schlichtkrull-l2':
            return self.scoring_function.s_penalty(batch, x)
        return self.scoring_function.relations.pow(2).sum()

    def encode(self, graph):
        x = F.relu(self.node_embeddings + self.node_embeddings_bias)
        x = self.rgc1(graph, features=x)
        if self.rgcn_layers == 2:
            x = self.rgc2(graph, features=F.relu(x))
        return x

    def forward(self, graph, triples):
        x = self.encode(graph)
        return self.scoring_function(triples, x), self.compute_penalty(triples, x)


class CompressionRelationPredictor(LinkPredictor):
    """c-rgcn: embeddings -> Linear(nemb, nhid) -> RGC layers at width nhid -> Linear(nhid, nemb) + residual."""

    def __init__(self, nnodes=None, nrel=None, nfeat=None, encoder_config=None, decoder_config=None):
        self._bottleneck = (encoder_config or {}).get("hidden1_size")
        super().__init__(nnodes, nrel, self._bottleneck, encoder_config, decoder_config)
        nemb = (encoder_config or {}).get("node_embedding")
        self.encoding_layer = nn.Linear(nemb, self._bottleneck)
        self.decoding_layer = nn.Linear(self._bottleneck, nemb)

    @staticmethod
    def _rgc1_in(nemb, nhid1):
        return nhid1  # upstream builds rgc1 with nemb inputs but feeds it nhid-wide features (F4c)

    def encode(self, graph):
        x = F.relu(self.node_embeddings + self.node_embeddings_bias)
        x = self.rgc1(graph, features=self.encoding_layer(x))
        if self.rgcn_layers == 2:
            x = self.rgc2(graph, features=F.relu(x))
        return self.node_embeddings + self.decoding_layer(x)
