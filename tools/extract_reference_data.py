"""Extract the price / capacity-factor series the goldens and the synthetic RTS-GMLC batches need.

Run ONCE in the build container (where /root/reference exists):

    python tools/extract_reference_data.py

Writes (data only, no reference source code):
  dispatches_amd/data/rts_gmlc_309.npz   8736 h of bus Carter/309:  LMP (RT), LMP DA, 309_WIND_1-RTCF, -DACF
        <- dispatches/case_studies/renewables_case/data/309_WIND_1-SimulationOutputs.csv
  dispatches_amd/data/rts_gmlc_303.npz   8784 h of bus Caesar/303 (same four columns)
        <- .../renewables_case/data/303_LMPs_15_reserve_500_shortfall.parquet
  dispatches_amd/data/nuclear_lmp_signal.npz  3100 x 24 day-signals in (set, year, cluster) order
        <- .../nuclear_case/lmp_signal.json
  dispatches_amd/data/price_taker_inputs.npz  inputs of the reference's price-taker design tests: 8760 hourly wind speeds
        <- .../renewables_case/data/44.21_-101.94_windtoolkit_2012_60min_80m.srw (column 3, m/s at 80 m; what PySAM's
        SRW_to_wind_data hands to tests/test_RE_flowsheet.py:33-34) and 8736 day-ahead LMPs
        <- .../renewables_case/tests/rts_results_all_prices.npy (second array, test_RE_flowsheet.py:25-27)
  dispatches_amd/data/nuclear_price_taker_lmps.npz  8784 hourly LMPs (real-time and day-ahead) at bus Attlee
        <- .../nuclear_case/report/rts_gmlc_15_500.csv (price_taker_analysis.py:45-113)
  tests/golden/reference_vectors.json    the known-answer vectors held by the reference's own tests and
        notebooks for this path, with file:line provenance.

Nothing under /root/reference is read at test / bench time; the GPU box only sees these files.
"""
import json
import os

import numpy as np
import pandas as pd

REF = "/root/reference/dispatches/case_studies/"
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = os.path.join(HERE, "dispatches_amd", "data")
    os.makedirs(out, exist_ok=True)

    df = pd.read_csv(REF + "renewables_case/data/309_WIND_1-SimulationOutputs.csv", index_col=0)
    np.savez_compressed(
        os.path.join(out, "rts_gmlc_309.npz"),
        start=np.array(str(df.index[0])),
        rt_lmp=df["LMP"].to_numpy(np.float64),
        da_lmp=df["LMP DA"].to_numpy(np.float64),
        rt_cf=df["309_WIND_1-RTCF"].to_numpy(np.float64),
        da_cf=df["309_WIND_1-DACF"].to_numpy(np.float64),
    )

    p = pd.read_parquet(REF + "renewables_case/data/303_LMPs_15_reserve_500_shortfall.parquet")
    np.savez_compressed(
        os.path.join(out, "rts_gmlc_303.npz"),
        start=np.array(str(p.index[0])),
        rt_lmp=p["LMP"].to_numpy(np.float64),
        da_lmp=p["LMP DA"].to_numpy(np.float64),
        rt_cf=p["303_WIND_1-RTCF"].to_numpy(np.float64),
        da_cf=p["303_WIND_1-DACF"].to_numpy(np.float64),
    )

    j = json.load(open(REF + "nuclear_case/lmp_signal.json"))
    days = []
    for s in sorted(k for k in j if k.isdigit()):
        for y in sorted(j[s], key=int):
            for c in sorted(j[s][y], key=int):
                d = j[s][y][c]
                days.append([d[str(h)] for h in range(1, 25)])
    np.savez_compressed(os.path.join(out, "nuclear_lmp_signal.npz"), lmp=np.asarray(days, np.float64))

    # ---- inputs of the price-taker design tests (test_RE_flowsheet.py:22-43) ----------------------------
    with open(REF + "renewables_case/tests/rts_results_all_prices.npy", "rb") as f:
        _ = np.load(f)
        da_lmp = np.load(f)
    srw = open(REF + "renewables_case/data/44.21_-101.94_windtoolkit_2012_60min_80m.srw").read().split("\n")
    speeds = np.array([float(line.split(",")[2]) for line in srw[5:5 + 8760]])
    assert da_lmp.shape == (8736,) and speeds.shape == (8760,)
    np.savez_compressed(os.path.join(out, "price_taker_inputs.npz"), wind_speed_m_s=speeds, da_lmp=np.asarray(da_lmp, np.float64))

    # ---- LMPs of the nuclear price-taker analysis (nuclear_case/report/price_taker_analysis.py:45-113) -----
    nd = pd.read_csv(REF + "nuclear_case/report/rts_gmlc_15_500.csv")
    assert len(nd) == 8784
    np.savez_compressed(os.path.join(out, "nuclear_price_taker_lmps.npz"), rt_lmp=nd["LMP"].to_numpy(np.float64),
                        da_lmp=nd["LMP DA"].to_numpy(np.float64))

    # ---- known-answer vectors of the reference's own tests / committed notebook outputs --------------
    golden = {
        "_provenance": "values transcribed from the reference's tests / notebook outputs (file:line below); "
                       "inputs are rows 0..47 of rts_gmlc_309.npz",
        "G1_self_schedule_p_max_mw": {
            "source": "dispatches/case_studies/renewables_case/tests/test_multiperiod_wind_battery_doubleloop.py:168-173",
            "reltol": 1e-2,
            "values": [0.0, 1.5734, 0.0, 0.0, 10.0865, 14.2151, 0.0, 0.0, 0.0, 0.0,
                       0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 26.8881, 1.3711, 4.7876,
                       20.5439, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
                       0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 86.0643, 0.0,
                       0.0, 0.0, 0.0, 0.0, 0.0, 35.7721],
        },
        "G2_bidder_last_point_cost": {
            "source": "dispatches/case_studies/renewables_case/tests/test_multiperiod_wind_battery_doubleloop.py:245-250",
            "reltol": 1e-2,
            "values": [0.0, 48.5758, 0.0, 0.0, 265.8715, 435.9852, 0.0, 0.0, 0.0,
                       0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1008.6438999999999,
                       49.4844, 161.6625, 550.2665999999999, 0.0, 0.0, 0.0, 0.0,
                       0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0,
                       0.0, 0.0, 0.0, 1623.0916, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 704.3113000000001],
        },
        "G3_tracker_wind_battery": {
            "source": "dispatches/case_studies/renewables_case/tests/test_multiperiod_wind_battery_doubleloop.py:75-111",
            "market_dispatch_mw": [0, 1.5, 15.0, 24.5],
            "expected_wind_power_kw": [1123.8, 1573.4, 20510.2, 25938.4],
            "rel": 1e-3, "abs": 1e-3,
        },
        "G3b_tracker_wind_pem": {
            "source": "dispatches/case_studies/renewables_case/tests/test_wind_PEM_double_loop.py:72-119",
            "market_dispatch_mw": [0, 1.5, 15.0, 24.5],
            "expected_wind_power_kw": [1123.85, 1573.38, 20510, 25938],
            "cap_factor0": 0.00562,
            "rel": 1e-3, "abs": 1e-3,
        },
        "G3c_pem_parametrized_da_p_max": {
            "source": "dispatches/case_studies/renewables_case/tests/test_wind_PEM_double_loop.py:152-158",
            "abstol": 1e-2,
            "values": [0.13, 1.08, 3.64, 15.37, 24.68, 31.83, 33.18, 13.89, 7.55, 4.99, 1.08,
                       0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.81, 1.89, 33.45, 34.12, 17.8, 13.49,
                       48.15, 69.72, 36.01, 111.67, 132.7, 178.56, 107.89, 146.86, 145.38, 143.9,
                       156.84, 111.94, 87.66, 154.69, 79.3, 116.12, 111.8, 98.18, 98.58, 106.54,
                       130.14, 124.75, 160.08, 153.47, 162.1],
        },
        "G3d_pem_parametrized_rt_last_cost": {
            "source": "dispatches/case_studies/renewables_case/tests/test_wind_PEM_double_loop.py:211-213",
            "reltol": 1e-2,
            "values": [33.72, 47.2, 615.31, 750.0],
        },
        "G4_nuclear_da_objective": {
            "source": "dispatches/case_studies/nuclear_case/nuclear_flowsheet_double_loop.ipynb:716 "
                      "(IPOPT objective, 3 identical scenarios x 48 h, min form) ; prices: cell 4",
            "ipopt_objective_3_scenarios": -1.6666883034970104e+06,
            "rel": 1e-6,
            "da_lmp": [21.288312, 20.419032, 19.689677, 19.983571, 19.983571, 20.419032,
                       21.843871, 23.437857, 18.072549, -0.0, -0.0, -0.0,
                       -0.0, -0.0, -0.0, -0.0, 18.861, 22.51634,
                       33.752674, 33.752674, 27.050323, 24.617429, 23.07, 19.689677],
            "rt_lmp": [23.07, 22.968387, 24.617429, 24.617429, 27.050323, 22.51634,
                       22.492903, 23.657742, 21.843871, -0.0, -0.0, -0.0,
                       -0.0, -0.0, -0.0, -0.0, -0.0, 21.916765,
                       24.079344, 23.40518, 22.683303, 21.916765, 20.244451, 18.861],
        },
        "G5_wind_battery_da_objective_xpress": {
            "source": "dispatches/case_studies/renewables_case/DoubleLoopOptimization.ipynb:657 "
                      "(Xpress, 3 identical scenarios x 28 h, max form; run with the OLDER constants "
                      "wind_op_cost=43, batt_rep_cost_kwh=150)",
            "xpress_objective_3_scenarios": -1.938255330232465e+04,
            "wind_mw": 147.6, "battery_mw": 25.0, "battery_mwh": 100.0, "horizon": 28,
            "wind_op_cost": 43.0, "batt_rep_cost_kwh": 150.0,
        },
        "G6_wind_battery_rt_objective_xpress": {
            "source": "dispatches/case_studies/renewables_case/DoubleLoopOptimization.ipynb:726,1139",
            "xpress_objective_3_scenarios_hour0": -6025.244239573585,
            "xpress_objective_3_scenarios_zero_price": -8694.246575342466,
            "horizon": 4,
        },
        "G7_battery_rows": {
            "source": "dispatches/unit_models/tests/test_battery.py:57-58,119",
            "case_a": {"elec_in": 5, "elec_out": 0, "soc0": 0, "e0": 0, "soc": 4.75, "throughput": 2.5},
            "case_b": {"soc0": 5, "e0": 5, "elec_out": 5, "soc": 0, "elec_in": 0.27701, "throughput": 7.6385},
        },
        "G8_price_taker_wind_battery": {
            "source": "dispatches/case_studies/renewables_case/tests/test_RE_flowsheet.py:123-133 (wind_battery_optimize, CBC; inputs "
                      ":22-43: LMPs capped at 200, wind speeds of the SRW file through PySAM's Weibull wind model)",
            "n_time_points": 168, "NPV": 666049365, "annual_revenue": 59163455, "rel": 1e-3,
            "battery_nameplate_power_kw": 0, "battery_abs": 1,
        },
        "G9_wind_unit_model": {
            "source": "dispatches/unit_models/tests/test_wind_power.py:49-50,78 (PySAM Windpower, ATB 5 MW turbine of "
                      "wind_power.py:128-143, 50 MW system, 10 m/s)",
            "speed_m_s": 10, "system_capacity_kw": 50000,
            "weibull_model_electricity_kw": 30083.39,          # resource_speed path (what the price-taker tests use)
            "distribution_model_capacity_factor": 0.5755, "distribution_model_electricity_kw": 28775.06, "rel": 1e-2,
        },
        "G10_price_taker_wind_battery_pem": {
            "source": "dispatches/case_studies/renewables_case/tests/test_RE_flowsheet.py:136-161 (wind_battery_pem_optimize, CBC, "
                      "6 x 24 periods, h2_price_per_kg = 2.5; first test design_opt = 'PEM' with batt_mw = 0, second the full design)",
            "time_points": 144, "h2_price_per_kg": 2.5,
            "batt_mw": 0, "pem_mw": 487, "annual_rev_h2": 155129116, "annual_rev_E": 68599396, "NPV": 1339462317, "rel": 1e-2,
            "pem_mw_abs_full_design": 1,
        },
        "constants": {
            "source": "dispatches/case_studies/renewables_case/load_parameters.py:24-79 + wind_battery_cost_parameter.json",
            "wind_op_cost": 41.78,
            "batt_rep_cost_kwh": 29.545625,
        },
    }
    # cross-check the constants against the json in the tree
    pj = json.load(open(REF + "renewables_case/wind_battery_cost_parameter.json"))
    assert abs(pj["battery"]["batt_cap_cost_param"]["moderate"]["2023"][0] * 0.5 / 4 - 29.545625) < 1e-12
    assert abs(pj["wind"]["fixed_om"]["moderate"]["2023"][0] - 41.78) < 1e-12
    with open(os.path.join(HERE, "tests", "golden", "reference_vectors.json"), "w") as f:
        json.dump(golden, f, indent=1)
    print("ok")


if __name__ == "__main__":
    main()
