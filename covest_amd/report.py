"""The step after the path: the result record of one estimate, field for field
what covest/data.py:106-173 print_output builds -- three likelihood evaluations (GPU, through the
model's compute_loglikelihood) and the genome size

    genome_size = round( sum_i i * h_i  /  correct_c(c * sample_factor) )        (covest/data.py:152-156)

The record is returned and, unless `silent`, printed as block-style YAML like the reference's output.
"""
import yaml

from . import __version__


def _none_filled(values, fallback):
    """`values` with its None entries taken from `fallback` (covest/data.py:94-103)."""
    if values is None or fallback is None:
        raise ValueError('Invalid arguments.')
    if len(values) != len(fallback):
        raise ValueError('Length of arguments should be equal.')
    return [f if v is None else v for v, f in zip(values, fallback)]


def _finite_int(x):
    return None if x == float('inf') else int(x)


def print_output(hist_orig, model, success, sample_factor, estimated=None, guess=None, orig=None,
                 reads_size=None, silent=False, orig_sample_factor=1, starting_points=1,
                 use_grid_search=False):
    def named(names, values):
        """{name: value} without the None entries; the coverage (first entry) is reported for the
        un-sampled data, i.e. times sample_factor."""
        if values is None or names is None:
            return {}
        row = [None if v is None else float(v) for v in values]
        if row[0] is not None and sample_factor is not None:
            row[0] *= sample_factor
        return {name: v for name, v in zip(names, row) if v is not None}

    record = {
        'model': model.short_name(),
        'hist_size': max(model.hist),
        'sample_factor': sample_factor,
        'orig_sample_factor': orig_sample_factor,
        'success': success,
        'version': __version__,
        'starting_points': starting_points,
        'use_grid_search': use_grid_search,
    }
    if guess is not None:
        record.update(named(('guessed_coverage', 'guessed_error_rate'), guess))
        record['guessed_loglikelihood'] = model.compute_loglikelihood(*guess)
    if estimated is not None:
        record.update(named(model.params, estimated))
        record['orig_coverage'] = float(estimated[0] * orig_sample_factor * sample_factor)
        record['loglikelihood'] = model.compute_loglikelihood(*estimated)
        occurrences = sum(i * n for i, n in hist_orig.items())
        record['genome_size'] = _finite_int(round(occurrences / model.correct_c(estimated[0] * sample_factor)))
        if reads_size is not None:
            record['genome_size_reads'] = _finite_int(
                round(reads_size / (estimated[0] * sample_factor * orig_sample_factor)))
    if orig is not None and any(orig):
        record.update(named(['provided_%s' % name for name in model.params], orig))
        try:
            record['provided_loglikelihood'] = model.compute_loglikelihood(*_none_filled(orig, estimated))
        except ValueError:
            pass
    if not silent:
        print(yaml.dump(record, indent=4, default_flow_style=False))
    return record
