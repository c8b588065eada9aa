"""Network spec of the hot path: the `net:` section of the reference's YAML -> dims and parameter table.

Mirrors the constructor arguments of the reference's ``SpeechRecognizer``
(lvsr/bricks/recognizer.py:176-204) and the Blocks parameter naming that is the checkpoint
contract (libs/blocks/blocks/model.py:60-161; names verified by instantiating the reference, see
oracle/theano_harness/gen_golden.py which asserts this table against the reference's own
``Model(cost).get_parameter_dict()``).

Pure Python + numpy on purpose: it is imported by the golden-vector generator that runs inside the
scratch Theano environment, by the tests and by the product.
"""
from collections import OrderedDict
import copy

SUPPORTED_ACTIVATIONS = ("maxout2", "rectifier", "tanh", "identity")

_DEFAULTS = dict(
    input_dim=None,            # F; reference: input_dims={'recordings': F}
    num_phonemes=None,         # V (number of labels incl. <eol>)
    eos_label=None,
    dims_bidir=None,           # [H]*n_layers
    subsample=None,            # [1]*n_layers if None (recognizer.py:237-238)
    dim_dec=None,              # D
    dec_stack=1,               # number of stacked decoder GRUs (RecurrentStack with skip connections, recognizer.py:250-262)
    dim_matcher=None,          # M; defaults to dim_dec (recognizer.py:224-225)
    attention_type="content",  # 'content' | 'content_and_conv'
    conv_n=None,               # c: filter half width
    conv_num_filters=1,        # K
    prior=None,                # dict(type=..., ...); None -> expanding 0/10000/0/0 (attention.py:74-77)
    energy_normalizer="softmax",
    post_merge_dims=None,      # [P] or None
    post_merge_activation="tanh",  # reference default Tanh() (recognizer.py:206-207)
    embed_outputs=True,
    dim_output_embedding=None,
    use_states_for_readout=True,
    data_prepend_eos=True,
    max_decoded_length_scale=1,
    bottom_dims=None,          # SpeechBottom(dims=...): MLP in front of the encoder (recognizer.py:105-126); None/[] = Identity
    bottom_activation="tanh",  # SpeechBottom default Tanh() (recognizer.py:116-117); the configs use Rectifier
)

DEFAULT_PRIOR = dict(type="expanding", initial_begin=0, initial_end=10000, min_speed=0, max_speed=0)


def normalize_net_config(cfg):
    """Fill defaults and validate the subset of the reference's `net:` schema this path supports."""
    out = copy.deepcopy(_DEFAULTS)
    for k, v in cfg.items():
        if k not in out:
            raise ValueError("unknown net config key %r" % (k,))
        out[k] = copy.deepcopy(v)
    for k in ("input_dim", "num_phonemes", "dims_bidir", "dim_dec"):
        if out[k] is None:
            raise ValueError("net config key %r is required" % (k,))
    out["dims_bidir"] = [int(d) for d in out["dims_bidir"]]
    if out["subsample"] is None:
        out["subsample"] = [1] * len(out["dims_bidir"])
    out["subsample"] = [int(s) for s in out["subsample"]]
    if len(out["subsample"]) != len(out["dims_bidir"]):
        raise ValueError("subsample and dims_bidir must have the same length")
    out["dec_stack"] = int(out["dec_stack"])
    if not 1 <= out["dec_stack"] <= 4:
        raise NotImplementedError("dec_stack must be between 1 and 4")
    if out["eos_label"] is None:
        out["eos_label"] = out["num_phonemes"] - 1
    if out["dim_matcher"] is None:
        out["dim_matcher"] = out["dim_dec"]
    if out["attention_type"] not in ("content", "content_and_conv"):
        # same error class as the reference (recognizer.py:275-277)
        raise ValueError("Unknown attention type {}".format(out["attention_type"]))
    if out["attention_type"] == "content_and_conv":
        if out["conv_n"] is None:
            raise ValueError("content_and_conv attention needs conv_n")
        if out["prior"] is None:
            out["prior"] = dict(DEFAULT_PRIOR)
        out["prior"].setdefault("type", "expanding")
    else:
        out["prior"] = None
    if out["energy_normalizer"] is None:
        out["energy_normalizer"] = "softmax"
    if out["energy_normalizer"] not in ("softmax", "logistic", "relu"):
        # same error as the reference (lvsr/bricks/attention.py:203-205)
        raise Exception("Unknown energey_normalizer: {}".format(out["energy_normalizer"]))
    if out["attention_type"] == "content" :
        out["energy_normalizer"] = "softmax"          # the Blocks content attention has no such option (recognizer.py:262-265)
    out["bottom_dims"] = [int(v) for v in (out["bottom_dims"] or [])]
    if out["bottom_dims"] and out["bottom_activation"] not in ("rectifier", "tanh", "identity"):
        raise NotImplementedError("bottom activation %r is not built" % out["bottom_activation"])
    if not out["post_merge_dims"]:
        out["post_merge_dims"] = None            # `post_merge_dims: []` (exp/wsj/configs/wsj_small.yaml) = no post-merge (recognizer.py:305 `if post_merge_dims:`)
    if out["post_merge_dims"] is not None:
        out["post_merge_dims"] = [int(v) for v in out["post_merge_dims"]]
        if len(out["post_merge_dims"]) > 4:
            raise NotImplementedError("post_merge_dims: at most 4 layers are built")
        if len(out["post_merge_dims"]) > 1 and out["post_merge_activation"] == "maxout2":
            # recognizer.py:305-319 builds MLP([act] * (n - 1) + [Identity()], [d // num_pieces for d in dims] + [V]): with Maxout
            # layer i emits d_{i+1} / pieces^2 values where layer i + 1 expects d_{i+1} / pieces (the source says so itself: "For
            # deeper Maxout network one has to use the Sequence brick") — the reference's own construction cannot run
            raise ValueError("post_merge_dims with more than one layer needs a one-piece activation (the reference's Maxout MLP "
                             "construction is inconsistent beyond one layer, lvsr/bricks/recognizer.py:305-319)")
        if out["post_merge_activation"] not in SUPPORTED_ACTIVATIONS:
            raise ValueError("post_merge_activation must be one of %s" % (SUPPORTED_ACTIVATIONS,))
        if out["post_merge_activation"] == "maxout2" and out["post_merge_dims"][0] % 2:
            raise ValueError("Maxout(2) needs an even post_merge dim")
    return out


class Dims(object):
    """Symbol table of SURVEY.md: F,H[],E,D,M,K,c,V,P,pieces,fb (feedback dim)."""

    def __init__(self, cfg):
        cfg = normalize_net_config(cfg)
        self.cfg = cfg
        self.F = cfg["input_dim"]
        self.bottom_dims = list(cfg["bottom_dims"])
        self.bottom_act = cfg["bottom_activation"]
        self.F_enc = self.bottom_dims[-1] if self.bottom_dims else self.F      # what the first BiGRU layer sees
        self.Hs = list(cfg["dims_bidir"])
        self.subsample = list(cfg["subsample"])
        self.n_layers = len(self.Hs)
        self.E = 2 * self.Hs[-1]
        self.D = cfg["dim_dec"]
        self.n_dec = cfg["dec_stack"]             # decoder GRU layers; the attention and the readout see all their states
        self.D_tot = self.n_dec * self.D
        self.M = cfg["dim_matcher"]
        self.V = cfg["num_phonemes"]
        self.conv = cfg["attention_type"] == "content_and_conv"
        self.K = cfg["conv_num_filters"] if self.conv else 0
        self.c = cfg["conv_n"] if self.conv else 0
        self.embed = bool(cfg["embed_outputs"])
        if self.embed:
            self.FB = self.D if cfg["dim_output_embedding"] is None else cfg["dim_output_embedding"]
        else:
            self.FB = self.V + 1          # OneOfNFeedback(num_phonemes + 1), recognizer.py:284
        if cfg["post_merge_dims"]:
            self.P = cfg["post_merge_dims"][0]
            self.act = cfg["post_merge_activation"]
            self.pieces = 2 if self.act == "maxout2" else 1
            self.Pout = self.P // self.pieces
            # further post-merge layers (one-piece activations only): widths of linear_0 .. linear_{n-2}; the last linear maps to V
            self.pm_hidden = [int(v) for v in cfg["post_merge_dims"][1:]]
        else:
            self.P = self.V
            self.act = "identity"
            self.pieces = 1
            self.Pout = self.V
            self.pm_hidden = []
        self.post_merge = bool(cfg["post_merge_dims"])
        self.use_states_for_readout = bool(cfg["use_states_for_readout"])
        self.normalizer = cfg["energy_normalizer"] if self.conv else "softmax"
        # ShallowEnergyComputer(use_bias = energy_normalizer != 'softmax') (lvsr/bricks/attention.py:66-69)
        self.energy_bias = self.normalizer != "softmax"

    def layer_input_dim(self, i):
        return self.F_enc if i == 0 else 2 * self.Hs[i - 1]

    def subsampled_length(self, T):
        for s in self.subsample:
            T = (T + s - 1) // s          # x[::s]
        return T


def decoder_layer_names(d, l):
    """Parameter names of decoder GRU layer `l` (0 = bottom).  One layer: the `transition` brick of AttentionRecurrent.  A stack
    (recognizer.py:250-262): RecurrentStack renames its children `transition_<l>#<l>`, exposes the sequences / states of layer
    l > 0 with the suffix `#<l>` (recurrent.py:786-798) — which the generator's Fork, the Distribute of the glimpse, the
    attention's state transformers and the readout's Merge all pick up as brick names — and feeds layer l from a bias-free
    Fork `fork_<l>` of the state of layer l - 1 (:823-831).  Names verified against the reference's own parameter dict
    (oracle/theano_harness/gen_golden.py)."""
    g = "/recognizer/generator"
    att = g + "/att_trans/" + ("conv_att" if d.conv else "cont_att")
    sfx = "" if l == 0 else "#%d" % l
    if d.n_dec == 1:
        trans = g + "/att_trans/transition"
    else:
        trans = g + "/att_trans/recurrentstack/transition_%d#%d" % (l, l)
    n = dict(Whh=trans + ".state_to_state", Whg=trans + ".state_to_gates", h0=trans + ".initial_state",
             Wdi=g + "/att_trans/distribute/fork_inputs%s.W" % sfx, Wdg=g + "/att_trans/distribute/fork_gate_inputs%s.W" % sfx,
             Wfi=g + "/fork/fork_inputs%s.W" % sfx, bfi=g + "/fork/fork_inputs%s.b" % sfx,
             Wfg=g + "/fork/fork_gate_inputs%s.W" % sfx, bfg=g + "/fork/fork_gate_inputs%s.b" % sfx,
             Ws=att + "/state_trans/transform_states%s.W" % sfx, Wms=g + "/readout/merge/transform_states%s.W" % sfx)
    if l > 0:
        n["Fi"] = g + "/att_trans/recurrentstack/fork_%d/fork_inputs.W" % l
        n["Fg"] = g + "/att_trans/recurrentstack/fork_%d/fork_gate_inputs.W" % l
    return n


def parameter_shapes(cfg):
    """name -> shape, names exactly as the reference's Model.get_parameter_dict()."""
    d = Dims(cfg)
    p = OrderedDict()
    for j, (i_, o_) in enumerate(zip([d.F] + d.bottom_dims[:-1], d.bottom_dims)):
        p["/recognizer/bottom/bottom/linear_%d.W" % j] = (i_, o_)           # MLP named "bottom" inside the bottom brick
        p["/recognizer/bottom/bottom/linear_%d.b" % j] = (o_,)
    for i, H in enumerate(d.Hs):
        I = d.layer_input_dim(i)
        for direction in ("forward", "backward"):
            base = "/recognizer/encoder/bidir%d/%s" % (i, direction)
            p[base + "/fork/fork_inputs.W"] = (I, H)
            p[base + "/fork/fork_inputs.b"] = (H,)
            p[base + "/fork/fork_gate_inputs.W"] = (I, 2 * H)
            p[base + "/fork/fork_gate_inputs.b"] = (2 * H,)
            p[base + "/gatedrecurrent.state_to_state"] = (H, H)
            p[base + "/gatedrecurrent.state_to_gates"] = (H, 2 * H)
            p[base + "/gatedrecurrent.initial_state"] = (H,)
    g = "/recognizer/generator"
    att = g + "/att_trans/" + ("conv_att" if d.conv else "cont_att")
    p[att + "/preprocess.W"] = (d.E, d.M)
    p[att + "/preprocess.b"] = (d.M,)
    for l in range(d.n_dec):
        p[decoder_layer_names(d, l)["Ws"]] = (d.D, d.M)
    p[att + "/energy_comp/linear.W"] = (d.M, 1)
    if d.energy_bias:
        p[att + "/energy_comp/linear.b"] = (1,)
    if d.conv:
        p[att + "/conv1d.filters"] = (d.K, 2 * d.c + 1)
        p[att + "/handler.W"] = (d.K, d.M)
    for l in range(d.n_dec):
        n = decoder_layer_names(d, l)
        p[n["Wdi"]] = (d.E, d.D)
        p[n["Wdg"]] = (d.E, 2 * d.D)
        if l > 0:       # fork of the state of the layer below, no bias with skip connections (recurrent.py:823-831)
            p[n["Fi"]] = (d.D, d.D)
            p[n["Fg"]] = (d.D, 2 * d.D)
        p[n["Whh"]] = (d.D, d.D)
        p[n["Whg"]] = (d.D, 2 * d.D)
        p[n["h0"]] = (d.D,)
    for l in range(d.n_dec):
        n = decoder_layer_names(d, l)
        p[n["Wfi"]] = (d.FB, d.D)
        p[n["bfi"]] = (d.D,)
        p[n["Wfg"]] = (d.FB, 2 * d.D)
        p[n["bfg"]] = (2 * d.D,)
    if d.use_states_for_readout:
        for l in range(d.n_dec):
            p[decoder_layer_names(d, l)["Wms"]] = (d.D, d.P)
    p[g + "/readout/merge/transform_weighted_averages.W"] = (d.E, d.P)
    if d.post_merge:
        p[g + "/readout/post_merge/bias.b"] = (d.P,)
        widths = [d.Pout] + d.pm_hidden + [d.V]                 # MLP([act] * (n - 1) + [Identity()], ...), recognizer.py:309-317
        for j in range(len(widths) - 1):
            p[g + "/readout/post_merge/mlp/linear_%d.W" % j] = (widths[j], widths[j + 1])
            p[g + "/readout/post_merge/mlp/linear_%d.b" % j] = (widths[j + 1],)
    else:
        p[g + "/readout/bias.b"] = (d.V,)
    if d.embed:
        p[g + "/readout/lookupfeedback/lookuptable.W"] = (d.V + 1, d.FB)
    return p


def count_parameters(cfg):
    n = 0
    for shape in parameter_shapes(cfg).values():
        k = 1
        for s in shape:
            k *= s
        n += k
    return n


# ----------------------------------------------------------------------------------------------
# Named configurations (BASELINE.json `configs`)
# ----------------------------------------------------------------------------------------------
def timit_tiny():
    return dict(input_dim=40, num_phonemes=62, dims_bidir=[128], subsample=[1], dim_dec=128, dim_matcher=128,
                attention_type="content", post_merge_dims=[128], post_merge_activation="maxout2",
                embed_outputs=True, data_prepend_eos=False)


def wsj_base(prior=None):
    return dict(input_dim=40, num_phonemes=33, dims_bidir=[256] * 4, subsample=[1, 1, 2, 2], dim_dec=256,
                dim_matcher=512, attention_type="content_and_conv", conv_n=100, conv_num_filters=10,
                prior=prior if prior is not None else dict(DEFAULT_PRIOR),
                post_merge_dims=[256], post_merge_activation="maxout2", embed_outputs=False,
                data_prepend_eos=False)


def wsj_deep():
    return dict(input_dim=40, num_phonemes=33, dims_bidir=[512] * 6, subsample=[1, 1, 1, 1, 2, 2], dim_dec=512,
                dim_matcher=512, attention_type="content_and_conv", conv_n=100, conv_num_filters=10,
                prior=dict(DEFAULT_PRIOR), post_merge_dims=[512], post_merge_activation="maxout2",
                embed_outputs=False, data_prepend_eos=False)


def wsj_paper():
    """The README-recommended model (exp/wsj/configs/wsj_paper7.yaml and its parent chain): 4 x 250 BiGRU on
    123-dimensional features (fbank + deltas), 250-unit decoder and matcher, ONE location filter of 201 taps, rectifier post-merge,
    embedded feedback, window_around_median(100, 100); trained with batch_size 10."""
    return dict(input_dim=123, num_phonemes=33, dims_bidir=[250] * 4, subsample=[1, 1, 2, 2], dim_dec=250, dim_matcher=250,
                attention_type="content_and_conv", conv_n=100, conv_num_filters=1,
                prior=dict(type="window_around_median", before=100, after=100),
                post_merge_dims=[250], post_merge_activation="rectifier", embed_outputs=True, data_prepend_eos=False,
                max_decoded_length_scale=3.0)


WORKLOADS = {
    # name: (net config factory, B, T, L)
    "timit_tiny": (timit_tiny, 2, 200, 40),
    "wsj_base": (wsj_base, 16, 800, 100),
    "wsj_deep": (wsj_deep, 8, 1500, 190),
    # the WSJ-base network with the two-layer RecurrentStack decoder of exp/wsj/configs/wsj_jan_wsj13v2.yaml (dec_stack: 2)
    "wsj_stack2": (lambda: dict(wsj_base(), dec_stack=2), 16, 800, 100),
    "wsj_paper": (wsj_paper, 10, 800, 100),
    # WSJ-base under the window prior of the full-size training fixture (tests/golden/wsj_base_median.npz): the clusters of the
    # persistent decoder exchange their windows' centres label by label
    "wsj_base_median": (lambda: wsj_base(prior=dict(type="window_around_median", before=10, after=100)), 16, 800, 100),
}


# ----------------------------------------------------------------------------------------------
# The reference's constructor keywords -> net config
# ----------------------------------------------------------------------------------------------
def _brick_name(obj):
    if obj is None:
        return None
    if isinstance(obj, str):
        return obj
    return getattr(obj, "__name__", None) or type(obj).__name__


def _activation_name(act):
    name = _brick_name(act)
    if name is None:
        return "tanh"                                  # recognizer.py:206-207
    low = name.lower()
    if low in SUPPORTED_ACTIVATIONS:
        return low
    if low == "maxout":
        pieces = getattr(act, "num_pieces", 2)
        if pieces != 2:
            raise NotImplementedError("Maxout(%d): only 2 pieces are built" % pieces)
        return "maxout2"
    raise NotImplementedError("post_merge_activation %r is not built" % name)


def from_reference_kwargs(input_dims=None, input_num_chars=None, eos_label=None, num_phonemes=None, dim_dec=None,
                          dims_bidir=None, enc_transition=None, dec_transition=None, use_states_for_readout=True,
                          attention_type="content", criterion=None, bottom=None, lm=None, character_map=None,
                          bidir=True, subsample=None, dims_top=None, prior=None, conv_n=None,
                          post_merge_activation=None, post_merge_dims=None, dim_matcher=None, embed_outputs=True,
                          dim_output_embedding=None, dec_stack=1, conv_num_filters=1, data_prepend_eos=True,
                          energy_normalizer=None, max_decoded_length_scale=1, name=None, **kwargs):
    """Map `SpeechRecognizer(**config['net'])` keywords (lvsr/bricks/recognizer.py:176-204) to the net config of
    this package.  Options whose bricks are not built raise NotImplementedError (never a silent fallback)."""
    if kwargs:
        raise TypeError("unknown SpeechRecognizer arguments: %s" % sorted(kwargs))
    for trans, which in ((enc_transition, "enc_transition"), (dec_transition, "dec_transition")):
        tn = _brick_name(trans)
        if tn is not None and tn != "GatedRecurrent":
            raise NotImplementedError("%s=%s: only GatedRecurrent is built" % (which, tn))
    if criterion is not None and dict(criterion).get("name", "log_likelihood") != "log_likelihood":
        raise ValueError("Unknown criterion {}".format(criterion["name"]))        # recognizer.py:296-297
    # `lm` / `character_map` do not change the network: SpeechRecognizer.__init__ builds the fusion model from them
    if not bidir:
        raise NotImplementedError("bidir=False is not built")
    if dims_top:
        # The reference cannot run this option either: it builds MLP([Tanh()], [dim_encoded] + dims_top + [dim_encoded])
        # (lvsr/bricks/recognizer.py:244-246) — ONE activation for len(dims_top) + 1 linear layers — and Blocks' MLP raises
        # ValueError when the two counts differ (libs/blocks/blocks/bricks/sequences.py:153-155).  Same error here.
        raise ValueError("dims_top: MLP with 1 activation and %d layers (the reference's own construction fails the same way)"
                         % (len(dims_top) + 1))
    bottom_dims, bottom_act = None, "tanh"
    if bottom is not None:
        bt = dict(bottom)
        bc = _brick_name(bt.get("bottom_class"))
        if bc is not None and bc != "SpeechBottom":
            raise NotImplementedError("bottom_class=%s: only SpeechBottom is built" % bc)
        bottom_dims = bt.get("dims") or None
        if bottom_dims:
            bottom_act = _activation_name(bt.get("activation"))
            if bottom_act == "maxout2":
                raise NotImplementedError("Maxout in the bottom MLP is not built")
    if isinstance(input_dims, dict):
        if list(input_dims) != ["recordings"]:
            raise NotImplementedError("only the 'recordings' input source is built")
        input_dim = input_dims["recordings"]
    else:
        input_dim = input_dims
    cfg = dict(input_dim=input_dim, num_phonemes=num_phonemes, eos_label=eos_label, dims_bidir=dims_bidir,
               subsample=subsample, dim_dec=dim_dec, dec_stack=dec_stack, dim_matcher=dim_matcher, attention_type=attention_type,
               conv_n=conv_n, conv_num_filters=conv_num_filters, prior=prior, energy_normalizer=energy_normalizer,
               post_merge_dims=post_merge_dims, embed_outputs=embed_outputs, dim_output_embedding=dim_output_embedding,
               use_states_for_readout=use_states_for_readout, data_prepend_eos=data_prepend_eos,
               max_decoded_length_scale=max_decoded_length_scale, bottom_dims=bottom_dims, bottom_activation=bottom_act)
    if post_merge_dims:
        cfg["post_merge_activation"] = _activation_name(post_merge_activation)
    return normalize_net_config(cfg)
