"""Configuration loader with the reference's semantics (lvsr/config.py:9-92): YAML files with `parent:` chains,
recursive merge, dotted command-line overrides parsed as YAML, `stages:` expanded in `number` order — and the
`!!python/name:` / `!!python/object/apply:` tags of the reference's configs resolved to `lvsr_amd.blocks_compat`
(a whitelist: arbitrary python objects are never constructed).  Validation follows the key set of
lvsr/configs/schema.yaml for the sections this path consumes.
"""
import copy
import os.path
from collections import OrderedDict

import yaml

from . import blocks_compat


class _Loader(yaml.SafeLoader):
    pass


def _resolve(path):
    try:
        return blocks_compat.REGISTRY[path]
    except KeyError:
        raise yaml.constructor.ConstructorError(
            None, None, "python object %r is not part of the attention-LVCSR hot path (known: %s)"
            % (path, ", ".join(sorted(blocks_compat.REGISTRY))), None)


def _name_constructor(loader, suffix, node):
    return _resolve(suffix)


def _apply_constructor(loader, suffix, node):
    cls = _resolve(suffix)
    if isinstance(node, yaml.SequenceNode):
        return cls(*loader.construct_sequence(node, deep=True))
    kw = loader.construct_mapping(node, deep=True)
    return cls(*kw.get("args", []), **kw.get("kwds", {}))


def _object_constructor(loader, suffix, node):
    """`!!python/object:X {attr: value}` (e.g. exp/wsj/configs/wsj_bhd11.yaml:5): state given as a mapping; the
    stand-ins take it as constructor keywords."""
    cls = _resolve(suffix)
    state = loader.construct_mapping(node, deep=True) if isinstance(node, yaml.MappingNode) else {}
    try:
        return cls(**state)
    except TypeError:
        obj = cls.__new__(cls)
        obj.__dict__.update(state)
        return obj


_Loader.add_multi_constructor("tag:yaml.org,2002:python/object:", _object_constructor)
_Loader.add_multi_constructor("tag:yaml.org,2002:python/name:", _name_constructor)
_Loader.add_multi_constructor("tag:yaml.org,2002:python/object/apply:", _apply_constructor)


def yaml_load(stream):
    return yaml.load(stream, Loader=_Loader)


def read_config(file_):
    """A YAML file with its `parent:` chain resolved (lvsr/config.py:9-21; parent paths may use $LVSR etc.): the oldest
    ancestor is the base, every descendant is merged over it in turn."""
    chain = []
    layer = yaml_load(file_)
    while True:
        chain.append(layer)
        parent = layer.get("parent")
        if parent is None:
            break
        with open(os.path.expandvars(parent)) as src:
            layer = yaml_load(src)
    merged = chain.pop()
    while chain:
        merge_recursively(merged, dict(chain.pop()))
    return merged


def merge_recursively(config, changes):
    """Overlay `changes` on `config` in place: mappings merge key by key, anything else replaces (lvsr/config.py:24-30)."""
    for key, new in changes.items():
        old = config.get(key)
        if isinstance(new, dict) and isinstance(old, dict):
            merge_recursively(old, new)
        else:
            config[key] = new


def make_config_changes(config, changes):
    """Command-line overrides (lvsr/config.py:33-49): `changes` = [("dotted.path", "yaml text"), ...]."""
    for dotted, text in changes:
        *parents, leaf = dotted.split(".")
        node = config
        for name in parents:
            node = node[name]
        node[leaf] = yaml_load(text) if isinstance(text, str) else text


# key sets of lvsr/configs/schema.yaml for the sections of this path
TOP_LEVEL = {"parent", "cmd_args", "data", "net", "initialization", "regularization", "training", "monitoring", "stages"}
NET_KEYS = {"bidir", "dim_dec", "dim_matcher", "dim_output_embedding", "dims_bidir", "post_merge_dims", "subsample",
            "dims_top", "dec_stack", "conv_n", "conv_num_filters", "enc_transition", "dec_transition", "attention_type",
            "use_states_for_readout", "max_decoded_length_scale", "criterion", "lm", "bottom", "post_merge_activation",
            "prior", "embed_outputs", "energy_normalizer", "data_prepend_eos", "character_map"}


class ConfigurationError(ValueError):
    pass


def validate(config):
    for k in config:
        if k not in TOP_LEVEL:
            raise ConfigurationError("unknown top-level section %r" % k)
    for k in config.get("net", {}):
        if k not in NET_KEYS:
            raise ConfigurationError("unknown key net.%s" % k)
    net = config.get("net", {})
    for k in ("dims_bidir", "subsample", "post_merge_dims", "dims_top"):
        if net.get(k) is not None and not (isinstance(net[k], list) and all(isinstance(v, int) for v in net[k])):
            raise ConfigurationError("net.%s must be a list of ints" % k)
    for k in ("dim_dec", "dim_matcher", "conv_n", "conv_num_filters", "dec_stack", "dim_output_embedding"):
        if net.get(k) is not None and not isinstance(net[k], int):
            raise ConfigurationError("net.%s must be an int" % k)


class Configuration(dict):
    """lvsr/config.py:52-92."""
    def __init__(self, config_path, schema_path=None, config_changes=(), validate_keys=True):
        """`schema_path` is accepted for signature compatibility; the key sets of lvsr/configs/schema.yaml are built in and
        checked unless `validate_keys=False` (the reference validates only when a schema path is given)."""
        with open(config_path, "rt") as src:
            config = read_config(src)
        make_config_changes(config, config_changes)
        self.multi_stage = "stages" in config
        if self.multi_stage:
            # stages with content, in `number` order; each is the whole configuration with the stage's changes merged in
            active = sorted(((name, ch) for name, ch in config["stages"].items() if ch), key=lambda item: item[1]["number"])
            self.ordered_stages = OrderedDict()
            for name, stage_changes in active:
                stage_changes.pop("number")       # in place, as lvsr/config.py:79 does: self["stages"][name] loses it too
                staged = copy.deepcopy({k: v for k, v in config.items() if k != "stages"})
                merge_recursively(staged, stage_changes)
                self.ordered_stages[name] = staged
        if validate_keys:
            validate(config)
            if self.multi_stage:
                for stage in self.ordered_stages.values():
                    validate(stage)
        super(Configuration, self).__init__(config)

    def net_kwargs(self, input_dim, num_phonemes, eos_label=None, stage=None, **extra):
        """The keyword set `SpeechRecognizer(**kw)` receives in lvsr/main.py:204-221 (create_model): the `net`
        section plus the data-derived sizes."""
        cfg = self.ordered_stages[stage] if stage is not None else self
        kw = copy.deepcopy(dict(cfg["net"]))
        kw.update(input_dims={"recordings": input_dim}, input_num_chars={}, num_phonemes=num_phonemes,
                  eos_label=num_phonemes - 1 if eos_label is None else eos_label)
        # create_model passes data.prepend_eos (lvsr/main.py:219); Data defaults it to False and asserts it is never True
        # (lvsr/datasets/__init__.py:163-166) — NOT the brick's own default of True
        kw.setdefault("data_prepend_eos", bool((cfg.get("data") or {}).get("prepend_eos", False)))
        kw.update(extra)
        return kw
