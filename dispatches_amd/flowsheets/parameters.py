"""Cost / size constants of the renewables and nuclear case studies.

Reference: dispatches/case_studies/renewables_case/load_parameters.py:24-79 (+ wind_battery_cost_parameter.json,
"moderate" 2023 column) and dispatches/case_studies/nuclear_case/nuclear_flowsheet.py:116-156,269-280,
nuclear_flowsheet_multiperiod_class.py:131-153.
"""
timestep_hrs = 1.0
h2_mols_per_kg = 500.0                      # load_parameters.py:26

# renewables
wind_op_cost = 41.78                        # $/kW-yr  fixed O&M
batt_cap_cost_kw = 236.365                  # $/kW  4-hr battery power-block cost
batt_rep_cost_kwh = batt_cap_cost_kw * 0.5 / 4      # 29.545625 $/kWh replacement cost (load_parameters.py:48)
pem_cap_cost = 1200.0
pem_op_cost = 0.03 * pem_cap_cost           # $/kW-yr
pem_var_cost = 0.0                          # $/kWh
battery_ramp_rate = 1e8                     # kWh per step (never binding)
battery_charging_eta = 0.95
battery_discharging_eta = 0.95
battery_degradation_rate = 1e-4
pem_electricity_to_mol = 0.00275984         # mol/s per kW (RE_flowsheet.py:131)

# nuclear
np_capacity_mw = 500.0
nuclear_pem_capacity_mw = 100.0
tank_capacity_kg = 5000.0
mw_h2 = 2.016e-3                            # kg/mol
nuclear_pem_electricity_to_mol = 0.002527406    # mol/s per kW
npp_vom = 2.3
nuclear_pem_vom = 1.3
tank_vom = 0.01
