"""Parser for the libpointmatcher ICP chain YAML the reference ships
(bruce_slam/config/icp.yaml:1-31, loaded by ``ICP.loadFromYaml``, pcl.cpp:187-197).

The file is accepted byte for byte.  Supported modules (anything else is rejected loudly --
silently ignoring a filter would change the pose):

    readingDataPointsFilters / referenceDataPointsFilters : must be empty
    matcher            KDTreeMatcher {knn: 1, epsilon: 0, maxDist}
    outlierFilters     MaxDistOutlierFilter {maxDist}, TrimmedDistOutlierFilter {ratio}
    errorMinimizer     PointToPointErrorMinimizer | PointToPlaneErrorMinimizer {force2D: 1}
    transformationCheckers  CounterTransformationChecker {maxIterationCount},
                            DifferentialTransformationChecker {minDiffRotErr, minDiffTransErr,
                                                               smoothLength}
    inspector          NullInspector
    logger             NullLogger
"""
import yaml

from ._lib import IcpParams


class IcpConfigError(ValueError):
    pass


def shipped_params(**over):
    """The chain of bruce_slam/config/icp.yaml as shipped."""
    p = dict(matcher_max_dist=10.0, use_max_dist_filter=1, max_dist_filter=3.0,
             use_trimmed_filter=1, trim_ratio=0.8, minimizer=0, max_iter=40, use_diff_checker=1,
             min_diff_rot=0.01, min_diff_trans=0.1, smooth_len=4, normals_knn=10)
    p.update(over)
    return IcpParams(**p)


def _single(node, what):
    """YAML module node -> (name, params dict).  Accepts 'Name' or {'Name': {...}}."""
    if isinstance(node, str):
        return node, {}
    if isinstance(node, dict) and len(node) == 1:
        (name, params), = node.items()
        return name, dict(params or {})
    raise IcpConfigError("cannot parse %s entry: %r" % (what, node))


def parse_icp_yaml(text):
    """YAML text -> IcpParams."""
    doc = yaml.safe_load(text) or {}
    known = {"readingDataPointsFilters", "referenceDataPointsFilters", "matcher", "outlierFilters",
             "errorMinimizer", "transformationCheckers", "inspector", "logger",
             "readingStepDataPointsFilters"}
    for key in doc:
        if key not in known:
            raise IcpConfigError("unknown ICP chain section %r" % key)
    for key in ("readingDataPointsFilters", "referenceDataPointsFilters", "readingStepDataPointsFilters"):
        if doc.get(key):
            raise IcpConfigError("%s are not supported (the shipped icp.yaml has none)" % key)

    p = dict(matcher_max_dist=float("inf"), use_max_dist_filter=0, max_dist_filter=0.0,
             use_trimmed_filter=0, trim_ratio=1.0, minimizer=0, max_iter=40, use_diff_checker=0,
             min_diff_rot=0.001, min_diff_trans=0.01, smooth_len=3, normals_knn=10)

    if "matcher" in doc and doc["matcher"] is not None:
        name, mp = _single(doc["matcher"], "matcher")
        if name != "KDTreeMatcher":
            raise IcpConfigError("unsupported matcher %r" % name)
        if int(mp.get("knn", 1)) != 1:
            raise IcpConfigError("KDTreeMatcher.knn must be 1")
        if float(mp.get("epsilon", 0)) != 0:
            raise IcpConfigError("KDTreeMatcher.epsilon must be 0 (exact search)")
        unknown = set(mp) - {"knn", "epsilon", "maxDist", "searchType"}
        if unknown:
            raise IcpConfigError("unsupported KDTreeMatcher parameters %r" % sorted(unknown))
        p["matcher_max_dist"] = float(mp.get("maxDist", float("inf")))

    for node in doc.get("outlierFilters") or []:
        name, fp = _single(node, "outlierFilters")
        if name == "MaxDistOutlierFilter":
            if p["use_max_dist_filter"]:
                raise IcpConfigError("MaxDistOutlierFilter listed twice")
            p["use_max_dist_filter"] = 1
            p["max_dist_filter"] = float(fp.get("maxDist", 1.0))
        elif name == "TrimmedDistOutlierFilter":
            if p["use_trimmed_filter"]:
                raise IcpConfigError("TrimmedDistOutlierFilter listed twice")
            p["use_trimmed_filter"] = 1
            p["trim_ratio"] = float(fp.get("ratio", 0.85))
        else:
            raise IcpConfigError("unsupported outlier filter %r" % name)

    if doc.get("errorMinimizer") is not None:
        name, ep = _single(doc["errorMinimizer"], "errorMinimizer")
        if name == "PointToPointErrorMinimizer":
            p["minimizer"] = 0
        elif name == "PointToPlaneErrorMinimizer":
            if int(ep.get("force2D", 0)) != 1:
                raise IcpConfigError("PointToPlaneErrorMinimizer needs force2D: 1 (clouds are 2-D)")
            p["minimizer"] = 1
        else:
            raise IcpConfigError("unsupported error minimizer %r" % name)

    for node in doc.get("transformationCheckers") or []:
        name, cp = _single(node, "transformationCheckers")
        if name == "CounterTransformationChecker":
            p["max_iter"] = int(cp.get("maxIterationCount", 40))
        elif name == "DifferentialTransformationChecker":
            p["use_diff_checker"] = 1
            p["min_diff_rot"] = float(cp.get("minDiffRotErr", 0.001))
            p["min_diff_trans"] = float(cp.get("minDiffTransErr", 0.001))
            p["smooth_len"] = int(cp.get("smoothLength", 3))
        else:
            raise IcpConfigError("unsupported transformation checker %r" % name)

    for key, ok in (("inspector", "NullInspector"), ("logger", "NullLogger")):
        if doc.get(key) is not None:
            name, _ = _single(doc[key], key)
            if name != ok:
                raise IcpConfigError("unsupported %s %r" % (key, name))
    return IcpParams(**p)
