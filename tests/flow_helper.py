"""TEST HELPER (not part of the package: the reference's `main`, covest/covest.py:99-195, is out of scope -- SURVEY.md 2
#4, "host Python stays as is" -- and a maintainer keeps calling it).  One estimate, start to finish, on
the GPU path -- `.hist` file -> process_histogram -> model -> first guess -> CoverageEstimator -> result record -- so that
tests/test_gpu_hist_steps.py::test_whole_default_flow can compare a whole run with the reference's recorded one."""
from pathlib import Path

from covest_amd import __version__, constants
from covest_amd.estimator import CoverageEstimator
from covest_amd.hist_steps import load_histogram, process_histogram, save_histogram
from covest_amd.models import select_model
from covest_amd.report import print_output


def estimate(input_histogram, kmer_size=constants.DEFAULT_K, read_length=constants.DEFAULT_READ_LENGTH,
             model='basic', trim=None, sample_factor=None, coverage=None, error_rate=None, params=(), fix=False,
             start_original=False, starting_points=1, grid=False, error_scale=constants.DEFAULT_ERR_SCALE,
             max_coverage=None, min_q1=constants.DEFAULT_MIN_SINGLECOPY_RATIO, reads_size=None, silent=True,
             rng=None, save_sampled=False, device=-1):
    """Returns the record print_output builds (and prints it unless `silent`).  `input_histogram` is a path or
    a {count: multiplicity} dict.  Keyword names follow the reference's command line options.  `device`: the HIP
    ordinal every step runs on (-1 = the calling thread's current device)."""
    if isinstance(input_histogram, dict):
        hist_orig, meta, stem = dict(input_histogram), {}, 'histogram'
    else:
        hist_orig, meta = load_histogram(input_histogram)
        stem = Path(input_histogram).stem
    hist, tail, sample_factor, guess_c, guess_e = process_histogram(
        hist_orig, kmer_size, read_length, trim=trim, sample_factor=sample_factor, rng=rng, device=device)
    orig_sample_factor = 1
    if 'sample_factor' in meta:
        try:
            orig_sample_factor = int(meta['sample_factor'])
        except ValueError as e:
            print(e)
    if sample_factor > 1 and save_sampled:
        save_histogram(hist, '%s.covest.sampled_x%d.hist' % (stem, sample_factor),
                       {'tool': 'covest_amd %s' % __version__, 'sample_factor': sample_factor * orig_sample_factor})
    if coverage:
        coverage /= sample_factor
    m = select_model(model)(kmer_size, read_length, hist, tail, max_error=constants.MAX_ERRORS,
                            max_cov=max_coverage, min_single_copy_ratio=min_q1, device=device)
    given = [None] * m.param_count
    for i, v in zip(range(m.param_count), (coverage, error_rate) + tuple(params)):
        given[i] = v
    pinned = given if fix else None
    if start_original:
        guess = list(given)
    else:
        guess = list(m.defaults)
        if not (guess_c == 0 and guess_e == 1):  # the moments gave a usable first guess
            guess[:2] = guess_c, guess_e
        if pinned:
            guess = [g if p is None else p for g, p in zip(guess, pinned)]
    if m.compute_loglikelihood(*guess) == -constants.INF:
        raise ValueError('Unable to compute likelihood. Please, try to trim the histogram, or use more complex model')
    est = CoverageEstimator(m, err_scale=error_scale, fix=pinned)
    res, success = est.compute_coverage(guess, starting_points=starting_points, use_grid_search=grid)
    return print_output(hist_orig, m, success, sample_factor, res, guess, given, reads_size=reads_size,
                        silent=silent, orig_sample_factor=orig_sample_factor, starting_points=starting_points,
                        use_grid_search=grid)
